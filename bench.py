#!/usr/bin/env python3
"""bench.py -- headline benchmark of the ray-tracing hot path on MI355X.

Metric (BASELINE.json): primary Mrays/s (+ frames/s) at 3840x2160, 1024 spheres,
1 spp, on 1/2/4/8 GPUs. One "step" = one complete frame with inputs and outputs
resident in HBM: every rank renders its rows with one HIP kernel launch (float4
linear-colour buffer + packed 0x00RRGGBB buffer), then -- for N > 1 -- a single
RCCL gather of the packed rows to rank 0 reassembles the image.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Rank 0 prints ONE JSON line. `value` comes from the strictly SERIAL loop at N = 1 (one
frame at a time on one stream; static camera, every table cached); everything else that
was measured in the same run is reported next to it under its own key:
  roofline         HBM roofline of the frame kernel: algorithmic bytes / the kernel's launch
                   duration (HIP events on the launch stream, serial loop) / 8 TB/s
  roofline_valu    the roof that binds (vector-instruction issue)
  moving_camera    the same loop with the camera nudged every frame, as the reference's
                   checkKey() does (kernel.cu:1716-1764): eye-cone table rebuilt per frame
  pipelined        two frames in flight on two streams (fill/drain of a launch overlapped)
  update_end_to_end  frames through rt_update(): kernel + D2H of the packed frame +
                   setPixelBuff memcpy, what the reference's boundary requires per frame
  cpu_baseline     the CPU oracle timed on this box's cores (all cores and one thread) on a
                   bounded row sample, with GPU-vs-CPU parity checked on those rows
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
EV_EVERY = 4                 # launches between two event-timed ones in the timed loop


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
    ap.add_argument("--tile-order", type=int, choices=(0, 1), default=1,
                    help="1 (library default): launches start their longest tiles first, from the wave durations of earlier "
                         "frames (rt_scene_set_tile_order); 0: grid order")
    ap.add_argument("--table-lds", action="store_true", help="stage the whole sphere table in LDS per workgroup")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the moving-camera / pipelined / update() legs")
    ap.add_argument("--no-preroll", action="store_true",
                    help="skip the untimed pre-roll that lets the GPU reach its clock under the workload before the timed region")
    ap.add_argument("--smi-clocks", type=int, choices=(0, 1), default=1,
                    help="1: fall back to rocm-smi (outside every timed region) when sysfs has no clock table")
    ap.add_argument("--cpu-band-stride", type=int, default=8,
                    help="CPU baseline renders every k-th 8-row band of the frame")
    ap.add_argument("--cpu-threads", type=int, default=16, help="host threads for the CPU baseline")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL over xGMI) for real runs; gloo + --share-gpu only to rehearse the "
                         "N>1 code path on a box with fewer GPUs than ranks")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: every rank uses cuda:0")
    ap.add_argument("--gather-words", type=int, default=24, choices=[24, 32],
                    help="N > 1: bits per pixel a rank sends to the root (24: packed word without its zero byte)")
    ap.add_argument("--frames-in-flight", type=int, default=0, choices=[0, 1, 2],
                    help="timed loop: 1 = strictly serial (default at N = 1), 2 = consecutive frames alternate "
                         "between two HIP streams with their own framebuffers (default at N > 1, where frame i's "
                         "gather overlaps frame i+1's kernel)")
    return ap.parse_args()


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rt, scene, args, gpu_rgba, gpu_packed):
    """Time the CPU oracle (kind 'port': the C restatement of the reference's
    kernel) on a deterministic row sample of the SAME workload -- with every host
    core, and with one thread on a smaller sample -- and check the GPU frame against it
    on those rows. bench.py's cpu_baseline leg is one of the three places allowed to touch oracle/."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = min(cores, args.cpu_threads)      # a 1-GPU box's CPU share is 16 cores
    w, h = args.width, args.height

    def run(bands, nthreads, check):
        rows = tests = mismatched = 0
        t_total = 0.0
        for y0 in bands:
            y1 = min(y0 + 8, h)
            t0 = time.perf_counter()
            rgba, packed, cnt = oracle_py.render(scene.spheres, scene.n_spheres, scene.texture, scene.sky, scene.sky_box,
                                                 scene.lights, scene.n_lights, rt.default_camera(), w, h,
                                                 rt.default_aspect(), y0=y0, y1=y1, nthreads=nthreads)
            t_total += time.perf_counter() - t0
            rows += y1 - y0
            tests += cnt["primary_tests"] + cnt["shadow_tests"]
            if check and gpu_rgba is not None:
                mismatched += int((rgba.view(np.uint32) != gpu_rgba[y0:y1].view(np.uint32)).any(axis=2).sum())
                mismatched += int((packed != gpu_packed[y0:y1]).sum())
        return rows, tests, mismatched, t_total

    bands = list(range(0, h, 8 * args.cpu_band_stride))
    rows, tests, mismatched, t_total = run(bands, cores, True)
    rays = rows * w
    one = bands[len(bands) // 2: len(bands) // 2 + 1]          # one 8-row band in the middle of the frame
    rows1, _, _, t1 = run(one, 1, False)
    return {
        "value": rays / t_total / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model(),
        "sample": f"every {args.cpu_band_stride}th 8-row band of the {w}x{h}/{args.spheres}-sphere frame "
                  f"({rows} rows, {rays} rays, {t_total:.1f} s)",
        "single_thread": {"value": rows1 * w / t1 / 1e6, "unit": "Mrays/s", "cores": 1,
                          "sample": f"rows {one[0]}..{one[0] + rows1 - 1} ({rows1 * w} rays, {t1:.1f} s)"},
        "tests_per_ray": tests / rays,
        "gpu_vs_cpu_mismatched_pixels": mismatched if gpu_rgba is not None else None,
    }


def committed_pmc(args, world):
    """Counters bench.py cannot collect itself, from the newest committed rocprofv3 PMC pass of this
    configuration (profiles/*_c3_frame_kernel_pmc_summary.json; FETCH_SIZE x2 + WRITE_SIZE as
    MI355X_MICROARCH.md prescribes). Applies to the default C3 single-GPU configuration only."""
    if world != 1 or (args.width, args.height, args.spheres, args.spp) != (3840, 2160, 1024, 1) or args.no_cull \
            or args.table_lds or args.tile not in (0, 8):
        return None
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_c3_frame_kernel_pmc_summary.json")))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            d = json.load(f)
        d["file"] = os.path.basename(files[-1])
        return d
    except Exception:
        return None


def read_clocks(fast_only=False):
    """Shader / memory clock of the GPU this process renders on, as the driver reports them at this moment
    ("record actual clocks with each run"). Called while launches are in flight, so that the
    reading is the clock under this workload and not the idle one. sysfs first (no subprocess), rocm-smi second;
    whatever cannot be read is None -- the bench never fails over it."""
    import glob
    import re
    import subprocess
    out = {"sclk_mhz": None, "mclk_mhz": None, "source": None}
    try:
        idx = torch.cuda.current_device()
        base = None
        # the device's own PCI function (a box shows every GPU of its host under /sys, whichever one is visible to HIP)
        pr = torch.cuda.get_device_properties(idx)
        if all(hasattr(pr, a) for a in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
            cand = "/sys/bus/pci/devices/%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if os.path.exists(os.path.join(cand, "pp_dpm_sclk")):
                base = cand
                out["pci"] = os.path.basename(cand)
        if base is None:
            cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/pp_dpm_sclk"),
                           key=lambda p: int(re.search(r"card(\d+)", p).group(1)))
            if idx < len(cards):       # the idx-th card that has a DPM table: right only when every GPU is visible
                base = os.path.dirname(cards[idx])
                out["pci"] = "unknown (card order)"
        if base is not None:

            def current(name):
                with open(os.path.join(base, name)) as f:
                    for line in f:
                        m = re.search(r"(\d+)\s*[Mm][Hh]z\s*\*", line)
                        if m:
                            return int(m.group(1))
                return None
            # the file reports one instantaneous reading of the power controller, and single readings scatter (a
            # loaded GPU was seen to answer 170, 1518 and 2409 MHz within one run): take several, report them all
            reads = [current("pp_dpm_sclk") for _ in range(3 if fast_only else 9)]
            reads = [r for r in reads if r is not None]
            if reads:
                out["sclk_mhz"] = max(reads)
                out["sclk_reads_mhz"] = reads
                out["mclk_mhz"] = current("pp_dpm_mclk")
                out["source"] = "sysfs pp_dpm_sclk/pp_dpm_mclk (level marked current; sclk = the highest of the readings listed)"
                return out
    except Exception:
        pass
    if fast_only:       # inside a timed region: never start a subprocess there
        return out
    try:
        r = subprocess.run(["rocm-smi", "-d", str(torch.cuda.current_device()), "--showclocks", "--json"],
                           capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout[r.stdout.index("{"):])
        card = next(iter(d.values()))
        for k, v in card.items():
            m = re.search(r"\((\d+)\s*[Mm][Hh]z\)", str(v))
            if not m:
                continue
            if "sclk" in k.lower() and out["sclk_mhz"] is None:
                out["sclk_mhz"] = int(m.group(1))
            if "mclk" in k.lower() and out["mclk_mhz"] is None:
                out["mclk_mhz"] = int(m.group(1))
        out["source"] = "rocm-smi --showclocks"
    except Exception as e:          # no rocm-smi, no permission, ...: say so
        out["source"] = f"unavailable ({type(e).__name__})"
    return out


def preroll(step, sync, reduce_max, window=10, tol=0.02, max_windows=60):
    """Untimed launches before the timed region until the GPU has reached its clock under this workload: windows
    of `window` frames are timed on the host until two consecutive ones agree within `tol` (a fresh box ramps its
    shader clock over the first few hundred launches; BENCH_r02's `value` loop ran first and was 11 % slow).
    Returns the step counter to continue from and what was observed."""
    k = 0
    hist = []
    for _ in range(max_windows):
        t0 = time.perf_counter()
        for _ in range(window):
            step(k, None)
            k += 1
        sync(k)
        hist.append(reduce_max((time.perf_counter() - t0) / window * 1e3)[0])   # every rank takes the same decision
        if len(hist) >= 3 and abs(hist[-1] - hist[-2]) <= tol * hist[-1] and abs(hist[-2] - hist[-3]) <= tol * hist[-2]:
            break
    return k, {"launches": k, "window": window, "tolerance": tol, "first_window_ms": hist[0], "last_window_ms": hist[-1],
               "windows": len(hist), "settled": len(hist) < max_windows}


def timed_loop(step, steps, warmup, sync, world, k0=0, mid=None):
    k = k0
    for _ in range(warmup):
        step(k, None)
        k += 1
    sync(k)
    t0 = time.perf_counter()
    for i in range(steps):
        step(k, i)
        k += 1
    side = mid() if mid else None      # (launches are asynchronous: this runs while the GPU works through them)
    sync(k)
    return time.perf_counter() - t0, k, side


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
    scene.set_tile_order(args.tile_order)
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
    nfl = args.frames_in_flight or (1 if world == 1 else 2)

    # outputs resident in HBM: float4 linear colour + packed words, band-local. Two sets: with two
    # frames in flight consecutive frames alternate between them (and between two streams).
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
    two_streams = [torch.cuda.Stream() for _ in range(2)]

    def make_fd(b, cam=None, fast=False):
        return scene.frame_desc(w, h, pixels=packed2[b].data_ptr(), rgba=rgba2[b].data_ptr(), y0=y0, y1=y1, spp=args.spp,
                                cull=not args.no_cull, tile=args.tile, interleave=interleave, cam=cam,
                                packed24=send2[b].data_ptr() if rgb24 else 0, table_lds=args.table_lds, fast=fast)

    fds = [make_fd(b) for b in range(2)]
    pending = [None, None]

    def finish(b, streams):
        """Complete frame b's gather and, on the root, put the rows in place (on b's stream)."""
        if pending[b] is not None:
            with torch.cuda.stream(streams[b]):
                pending[b].wait()
                pending[b] = None
                if rank == 0:
                    nonlocal frame
                    frame = roots[b].assemble()       # one index_select puts every row in place

    gather_end = [None, None]     # root: event to record once buffer set b's gather and assembly are on its stream

    def make_step(streams, ev, fd_of, ev_g=None):
        def step(k, i):
            b = k & 1
            if world > 1:
                finish(b, streams)             # buffer set b is free again
                if gather_end[b] is not None:  # ... and its exchange (gather + assembly on the root) ends here in stream order
                    gather_end[b].record(streams[b])
                    gather_end[b] = None
            with torch.cuda.stream(streams[b]):
                timed = i is not None and ev is not None and i % EV_EVERY == 0
                if timed:
                    ev[i // EV_EVERY][0].record(streams[b])
                scene.render_raw(fd_of(k, b), streams[b].cuda_stream)
                if timed:
                    ev[i // EV_EVERY][1].record(streams[b])
                    if ev_g is not None:       # the frame's exchange: from the end of this rank's kernel ...
                        ev_g[i // EV_EVERY][0].record(streams[b])
                        gather_end[b] = ev_g[i // EV_EVERY][1]
                if world > 1:                  # the frame's single collective, asynchronous, ordered after b's kernel
                    src = send2[b] if coll_dev == "cuda" else send2[b].cpu()
                    if rank == 0:
                        pending[b] = dist.gather(src, gathered2[b], dst=0, async_op=True)
                    else:
                        pending[b] = dist.gather(src, None, dst=0, async_op=True)
        return step

    def make_sync(streams):
        def sync(k):
            if world > 1:
                for b in (k & 1, (k + 1) & 1):
                    finish(b, streams)
                    if gather_end[b] is not None:
                        gather_end[b].record(streams[b])
                        gather_end[b] = None
            torch.cuda.synchronize()
            if world > 1:
                dist.barrier()
        return sync

    def reduce_max(*vals):
        t = torch.tensor(list(vals), dtype=torch.float64, device=coll_dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(v) for v in t]

    # ---------------- the timed loop the contract asks for ----------------
    streams = two_streams if nfl == 2 else [main_stream, main_stream]
    # HIP events around every EV_EVERY-th launch of the timed loop (a pair of event records costs the
    # loop ~15 us of stream time, 3 % of a frame: sampling keeps the timed region honest)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range((args.steps + EV_EVERY - 1) // EV_EVERY)]
    ev_g = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ev] if world > 1 else None
    step_main = make_step(streams, ev, lambda k, b: fds[b], ev_g)
    sync_main = make_sync(streams)
    # untimed pre-roll until the clock has settled under this workload (the timed region stays `steps` frames after
    # `warmup` more); clocks read while launches are in flight, before and at the end of the timed loop
    clocks = {}
    k0 = 0
    if not args.no_preroll:
        for _ in range(50):
            step_main(k0, None)
            k0 += 1
        if rank == 0:
            clocks["before"] = read_clocks(fast_only=not args.smi_clocks)
        sync_main(k0)
    # (the pre-roll comes AFTER the clock reading: a query of the power controller disturbs the launches around it)
    pre = None
    if not args.no_preroll:
        kpre, pre = preroll(lambda k, i: step_main(k0 + k, i), lambda k: sync_main(k0 + k), reduce_max)
        k0 += kpre
    # (no clock reading inside the timed region: a read of the power controller's tables takes a millisecond or two and was
    # measured to stretch a 20-step loop by 17 %; the clocks are read during untimed launches right before and right after)
    elapsed, k, _ = timed_loop(step_main, args.steps, args.warmup, sync_main, world, k0=k0)
    kk = k
    for _ in range(50):
        step_main(kk, None)
        kk += 1
    if rank == 0:
        clocks["after"] = read_clocks(fast_only=True)
    sync_main(kk)
    packed = packed2[(kk - 1) & 1]
    rgba = rgba2[(kk - 1) & 1]
    k = kk
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    gather_ms = float(np.mean([a.elapsed_time(b) for a, b in ev_g])) if ev_g else 0.0
    # per-rank figures ("report per-GPU kernel times"): every rank's own launch duration and loop time
    per_rank = None
    if world > 1:
        mine = torch.tensor([kernel_ms, elapsed / args.steps * 1e3, float(rows)], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"kernel_ms": [float(t[0]) for t in allr], "ms_per_step": [float(t[1]) for t in allr],
                    "rows": [int(t[2]) for t in allr]}
    elapsed, kernel_ms = reduce_max(elapsed, kernel_ms)
    ms_per_step = elapsed / args.steps * 1e3

    extras = {}
    if not args.no_extras and world == 1:
        # camera nudged every frame (kernel.cu:1727 `cam.Org.z += 0.1`, back and forth so the view stays
        # the workload's): the eye-cone table is rebuilt on the device for every frame
        def cam_at(j):                 # eight positions 0.1 apart, walked back and forth
            cam = rt.default_camera()
            cam.Org.z = 10.0 + 0.1 * (j - 4)
            return cam

        def cam_of(k):
            return cam_at(k % 8 if (k // 8) % 2 == 0 else 7 - (k % 8))
        s1 = [main_stream, main_stream]
        e_mv, _, _ = timed_loop(make_step(s1, None, lambda k, b: make_fd(b, cam_of(k))), args.steps, args.warmup, make_sync(s1), world)
        # the same positions, each held for an eighth of the loop: the same pixels to render, 8 table builds in all
        hold = max(1, (args.steps + args.warmup + 7) // 8)
        e_st, _, _ = timed_loop(make_step(s1, None, lambda k, b: make_fd(b, cam_at((k // hold) % 8))), args.steps, args.warmup,
                                make_sync(s1), world)
        e_pl, _, _ = timed_loop(make_step(two_streams, None, lambda k, b: fds[b]), args.steps, args.warmup, make_sync(two_streams), world)
        # the launch-order A/B under IDENTICAL conditions: the same serial loop, the same sampled HIP events, the same
        # steps and warm-up, back to back on the settled GPU -- first with the tiles started in grid order (what a launch
        # does without durations of earlier frames), then once more in the library's order
        ev_ab = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ev]
        scene.set_tile_order(0)
        e_go, _, _ = timed_loop(make_step(s1, ev_ab, lambda k, b: fds[b]), args.steps, args.warmup, make_sync(s1), world)
        scene.set_tile_order(args.tile_order)
        e_lo, _, _ = timed_loop(make_step(s1, ev_ab, lambda k, b: fds[b]), args.steps, args.warmup, make_sync(s1), world)
        e_mv, e_st, e_pl, e_go, e_lo = reduce_max(e_mv, e_st, e_pl, e_go, e_lo)
        rays = w * h * args.spp
        extras["grid_order"] = {"ms_per_step": e_go / args.steps * 1e3, "Mrays_per_s": rays / (e_go / args.steps) / 1e6,
                                "library_order_same_conditions_ms": e_lo / args.steps * 1e3,
                                "gain_of_the_library_order": 1.0 - e_lo / e_go,
                                "note": "rt_scene_set_tile_order(0): tiles started row-major as the grid comes. The default starts blocks "
                                        "of 16x16 tiles in the order of their longest tile (wave durations recorded by the frame kernel in "
                                        "the previous launch, sorted on the device after 1, 2, 4, 8, ... launches of an unchanged view and "
                                        "every third launch of a moving camera) so that a launch does not drain behind a few expensive tiles"}
        extras["moving_camera"] = {"ms_per_step": e_mv / args.steps * 1e3, "Mrays_per_s": rays / (e_mv / args.steps) / 1e6,
                                   "same_positions_held_ms": e_st / args.steps * 1e3,
                                   "overhead_vs_held": e_mv / e_st - 1.0,
                                   "note": "one frame at a time, cam.Org.z nudged by 0.1 every frame (kernel.cu:1727); eye-cone table "
                                           "rebuilt by a device kernel on the frame's stream, tile order re-sorted every third frame, "
                                           "no host synchronisation"}
        extras["pipelined"] = {"frames_in_flight": 2, "ms_per_step": e_pl / args.steps * 1e3,
                               "Mrays_per_s": rays / (e_pl / args.steps) / 1e6,
                               "note": "consecutive frames alternate between two HIP streams with their own framebuffers"}
        # restore the last static frame in buffer set of `packed` for the parity check below
        scene.render_raw(fds[(k - 1) & 1], main_stream.cuda_stream)
        torch.cuda.synchronize()
        # the opt-in approximate mode (rt_launch_opts.fast), NOT the metric: same loop, and its frame against the exact one
        if not args.no_cull and not args.table_lds and args.tile in (0, 8):
            exact = rgba2[(k - 1) & 1].clone()
            fds_fast = [make_fd(b, fast=True) for b in range(2)]
            e_fa, kf, _ = timed_loop(make_step(s1, None, lambda k, b: fds_fast[b]), args.steps, args.warmup, make_sync(s1), world)
            got = rgba2[(kf - 1) & 1]
            rel = (got[..., :3].double() - exact[..., :3].double()).abs() / exact[..., :3].double().abs().clamp_min(1e-3)
            worst = rel.amax(dim=2)
            flipped = worst > 1e-5
            extras["fast_mode"] = {"opt_in": "rt_launch_opts.fast = 1; never the default, never `value`",
                                   "ms_per_step": e_fa / args.steps * 1e3, "Mrays_per_s": rays / (e_fa / args.steps) / 1e6,
                                   "flipped_pixel_fraction": float(flipped.double().mean()),
                                   "max_rel_error_on_non_flipped_pixels": float(worst[~flipped].max()) if bool((~flipped).any()) else None,
                                   "bit_identical_pixel_fraction": float((worst == 0).double().mean()),
                                   "note": "relative error per channel of the float4 frame against the exact frame of this run "
                                           "(= the oracle bit for bit); flipped = beyond 1e-5: a shadow sample or a texel decided the other way"}
            scene.render_raw(fds[(k - 1) & 1], main_stream.cuda_stream)
            torch.cuda.synchronize()

    # executed-work statistics from the instrumented kernel variant (untimed)
    stats = scene.render(w, h, y0=y0, y1=y1, want_stats=True, spp=args.spp, cull=not args.no_cull, tile=args.tile,
                         interleave=interleave)["stats"]
    st = torch.tensor([stats[k] for k in rt.STAT_NAMES], dtype=torch.float64, device=coll_dev)
    if world > 1:
        dist.all_reduce(st, op=dist.ReduceOp.SUM)
    stats = dict(zip(rt.STAT_NAMES, [float(v) for v in st.cpu()]))

    if rank == 0:
        rays = w * h * args.spp
        value = rays / (ms_per_step * 1e-3) / 1e6
        # algorithmic bytes of one launch: float4 + packed word (+ 24-bit copy for the gather) per pixel-sample
        band_bytes = rows * w * (16 + 4 + (3 if rgb24 else 0)) * 1.0
        # the launch's duration: HIP events on its stream. With two frames in flight an event pair spans two
        # overlapped kernels, so the per-frame time is the honest denominator there.
        launch_ms = kernel_ms if nfl == 1 else ms_per_step
        achieved = band_bytes / (launch_ms * 1e-3) / 1e9
        slots = stats["wave_test_slots"] + stats["cull_tests"]     # lane slots issued for sphere/beam tests
        valu_tflops = slots * FLOP_PER_TEST / (ms_per_step * 1e-3) / 1e12
        pmc = committed_pmc(args, world)
        valu = {}
        if pmc:
            try:
                per_wave = pmc["derived"]["valu_insts_per_wave"]
                waves = pmc["pmc_avg_per_dispatch"]["SQ_WAVES"]
                rate = per_wave * waves / (256 * 4) / (launch_ms * 1e-3)      # VALU instructions / s / SIMD
                valu = {"valu_insts_per_wave_pmc": per_wave, "salu_insts_per_wave_pmc": pmc["derived"].get("salu_insts_per_wave"),
                        "pmc_file": pmc["file"],
                        "valu_issue_rate_per_simd_Ginst_s": rate / 1e9, "valu_issue_nominal_per_simd_Ginst_s": 1.2,
                        "valu_issue_frac_of_nominal": rate / 1.2e9,
                        "note_issue": "nominal = one wave64 instruction per 2 cycles per SIMD at 2.4 GHz; plain fp32 ops reach "
                                      "0.95 G/s on this chip, compares/selects/binary64 0.55, transcendentals 0.29 "
                                      "(profiles/r01_valu_issue_rates.txt)"}
            except Exception:
                valu = {}
        out = {
            "metric": "primary_Mrays_per_s", "value": value, "unit": "Mrays/s",
            "frames_per_s": 1e3 / ms_per_step,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"C3: {w}x{h}, {args.spheres} spheres (MSVC rand() replay seed {args.seed}), "
                                   f"{args.spp} spp, 3 lights x 10 shadow samples; static camera, one wave64 per 8x8 tile, "
                                   f"each tile's culled sphere lists staged in LDS and broadcast across lanes "
                                   + ("(whole table staged in LDS per workgroup)" if args.table_lds else
                                      "(the 16 KiB tables themselves are read from L2; opts.table_lds stages them whole)"),
                       "cull": not args.no_cull, "tile": args.tile or 8,
                       "tile_order": "blocks of 16x16 tiles, longest tile first, from the wave durations of the previous launches "
                                     "(scheduling only; rt_scene_set_tile_order); see grid_order" if args.tile_order else "grid order",
                       "parallelism": f"{BLOCK}-row blocks round-robin x{world} + 1 RCCL gather per frame of "
                                      f"{24 if rgb24 else 32}-bit pixels (overlapped with the next frame's kernel)"
                                      if world > 1 else "single GPU",
                       "frames_in_flight": nfl,
                       "outputs": "float4 RGBA + packed 0x00RRGGBB in HBM"},
            "kernel_ms": kernel_ms,
            "clocks": {**clocks, "nominal_sclk_mhz": 2400,
                       "note": "read during 50 untimed launches of this workload right before and right after the timed region "
                               "(an idle GPU reports its idle clock; a read inside the region would stretch it); "
                               "every fraction of a peak in this line is against the NOMINAL peak (8 TB/s, 157.3 TFLOP/s, "
                               "2.4 GHz issue), not scaled to the clock read here"},
            "preroll": pre,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc["derived"].get("hbm_traffic_bytes") if pmc else None,
                         "algorithmic_bytes_per_launch": band_bytes, "launch_ms": launch_ms,
                         "note": "achieved = algorithmic_bytes_per_launch / launch_ms; launch_ms = mean HIP-event duration of the "
                                 "frame kernel on its stream in this run's serial loop"
                                 + ("" if nfl == 1 else " (two frames in flight: time per frame instead)")
                                 + ". HBM is NOT the binding roof for this path (SURVEY.md F2); see roofline_valu."},
            "roofline_valu": {"bound": "valu_fp32", "achieved": valu_tflops, "peak": VALU_PEAK_TFLOPS,
                              "unit": "TFLOP/s", "frac": valu_tflops / VALU_PEAK_TFLOPS,
                              "executed_test_slots": slots,
                              "note": "64-lane issue slots of sphere + beam tests x 17 flop / frame time; "
                                      "excludes ray setup and shading ALU work (most of the instructions)",
                              **valu},
            "work": {"hit_fraction": stats["hit_pixels"] / rays, "primary_lane_tests_per_ray": stats["primary_tests"] / rays,
                     "shadow_lane_tests_per_ray": stats["shadow_tests"] / rays,
                     "cull_tests_per_ray": stats["cull_tests"] / rays,
                     "list_overflows": stats["list_overflows"]},
            **extras,
        }
        if world == 1 and not args.no_extras and (w, h, args.spp) == (3840, 2160, 1) and not args.no_cull:
            out["update_end_to_end"] = update_end_to_end(rt, args, packed[:rows])
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
            out["per_rank"] = per_rank
            out["gather_ms"] = {"root_kernel_end_to_assembled_ms": gather_ms,
                                "note": "HIP events on the root's frame stream around every 4th frame's exchange: from the end of "
                                        "the root's own kernel to the gather having delivered every rank's rows and "
                                        "rt_assemble_rows24 having put them in place (includes waiting for the slowest rank)"}
            # the assembled frame against a full single-GPU render of the same frame (untimed, rank 0)
            full = scene.render(w, h, spp=args.spp, cull=not args.no_cull, tile=args.tile, want_rgba=False)["packed"]
            torch.cuda.synchronize()
            out["gathered_frame_shape"] = list(frame.shape)
            out["gathered_frame_equals_single_gpu_render"] = bool(torch.equal(frame.to(full.device), full))
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def update_end_to_end(rt, args, packed_ref):
    """Frames through the reference's own boundary: rt_update() = size query -> launch(es) -> D2H of the packed
    frame into pinned memory -> setPixelBuff memcpy into the offscreen window (kernel.cu:1762-1792)."""
    import ctypes as C
    lib = rt.load_library()
    lib.rt_config_set_sphere_count(args.spheres)
    lib.rt_config_set_seed(args.seed)
    lib.rt_on_start()
    lib.rt_offscreen_resize(args.width, args.height)
    n = max(5, min(30, args.steps))
    for _ in range(3):
        lib.rt_update()
    t0 = time.perf_counter()
    for _ in range(n):
        lib.rt_update()
    ms = (time.perf_counter() - t0) / n * 1e3
    got = np.ctypeslib.as_array(lib.rt_offscreen_pixels(), shape=(args.height, args.width))
    same = bool(np.array_equal(got.view(np.uint32), packed_ref.cpu().numpy().view(np.uint32)))
    # the same with the camera moved before every frame
    cam = lib.rt_config_camera()
    t0 = time.perf_counter()
    for i in range(n):
        cam.contents.Org.z = 10.0 + (0.1 if i % 2 else 0.0)
        lib.rt_update()
    ms_mv = (time.perf_counter() - t0) / n * 1e3
    cam.contents.Org.z = 10.0
    return {"ms_per_frame": ms, "frames_per_s": 1e3 / ms, "frames": n, "presented_frame_equals_kernel_frame": same,
            "moving_camera_ms_per_frame": ms_mv,
            "device_ms_last_frame": float(lib.rt_last_frame_ms()),
            "note": "host wall clock around rt_update(): kernel in 3 row bands + D2H of 33 MB into pinned memory "
                    "(overlapped band by band) + setPixelBuff memcpy; PCIe-inclusive, never the bench `value`"}


if __name__ == "__main__":
    main()
