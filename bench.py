#!/usr/bin/env python3
"""bench.py - Mpixels/s of the fused remap on BASELINE.json's configs, with the roofline of its kernel.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config c2|c1|c3|c5|c4shard|c5shard] [--batch B]

``--gpus N`` with N > 1 and no RANK in the environment makes this script START the N ranks itself
(``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...``) before
anything touches a GPU, wait for them and exit with their return code; under an external launcher it joins
the group it is given (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  A rank count that differs from --gpus is an
error, never a silent 1-GPU line.

A *step* is one launch of the hot path (``pb_remap_u8``) over B frames (B = 1 unless --batch or a *shard*
config says otherwise).  Configs (BASELINE.json):
    c2       (default, the headline) one 8192x4096 equirect frame -> 4096x4096 equidistant-360 inscribed
    c1       3072x3072 equidistant-360 fisheye -> 4096x2048 equirect
    c3       4096x4096 equidistant-360 -> equisolid-360 with rotation (30, 45, 10) degrees
    c5       7776x3888 double fisheye (2 x 180 degrees) -> 8192x4096 equirect stitch
    c4shard  one GPU's share of BASELINE config 4: 64 distinct c2 frames resident, B frames per launch (default 8)
    c5shard  one GPU's share of BASELINE config 5: 32 distinct c5 frames resident, B frames per launch (default 8)
Every rank owns a pool of distinct synthetic frames resident in HBM and cycles through it, so the timed
launches stream from HBM rather than re-reading one frame out of the 256 MiB Infinity Cache.  Ranks are
independent (frames shard, nothing pixel-sized crosses xGMI); the only collective is the RCCL broadcast of the
parameter block from rank 0 and the barriers / max-reduce that bracket the timed region.  Rank 0 prints ONE
JSON line.

At N = 1 the line also carries ``configs``: c1, c3, c5, c4shard and c5shard measured in the SAME process right after the
headline (plan, frame pool beyond the Infinity Cache, HIP-event timed launches, algorithmic / must-move bytes from the plan's own
index map checked against the reference's figure), plus ``single_image_ms`` (a new geometry's warm plan + first frame) and
``faithful_kernel_ms`` (the float64 chain, what a deferred plan runs) for the headline config - so that one driver-run line shows
every kernel the design document talks about (cross-check: profiles/r03_*_kernel_stats.csv).

``roofline`` prices the kernel's ALGORITHMIC bytes (3 B written per output pixel + 3 B read per in-bounds source
sample; recomputed here from the plan's own index map and checked against tests/golden/full.json, which holds
the reference's figure) against the 8 TB/s HBM3E peak, using per-launch HIP-event durations taken on the launch
stream inside the timed region.  Next to it: ``copy_ceiling_gbs`` (a plain device copy measured in this run),
``attainable_frac`` (algorithmic bytes / the bytes that MUST cross HBM: every 128-byte source line that holds a
sample + the output), p10 / p90 of the per-launch durations, ``plan_create_ms`` (cold: the process's first plan, code
object load included), ``plan_create_warm_ms`` and ``first_frame_ms`` (the once-per-geometry cost that precedes the
timed region).  ``cpu_baseline`` times the NumPy oracle (the pinned
restatement of the reference path) on this host, whole path and gather-only (coordinate map cached).
"""

from __future__ import annotations

import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
# The frames a rank rotates through must be COLD: 1.25 GiB, five times the 256 MB Infinity Cache.  With 1.4 x (round 1-3's 320 MB
# rule in the `configs` block) the cache still served part of every frame: c1 12.3 us instead of 16.2, c3 29.5 instead of 33
# (experiments/session_r3_ai.sh; c2 and c5 were not affected).
POOL_BYTES_MIN = 1280 << 20

# name -> geometry (degrees; magnitude as the CLI computes it), pool = frames resident per GPU,
# pin = the entry of tests/golden/full.json that holds the reference's algorithmic bytes for the geometry
CONFIGS = {
    "c1": dict(dst=("pano", 2048, 4096), src=("camera", 3072, 3072, "equidistant", 360.0, 1535.5), rot=[], mask=1, pool=8, batch=1, pin="c1",
               text="c1: one 3072x3072 equidistant-360 fisheye frame -> 4096x2048 equirectangular per step"),
    "c2": dict(dst=("camera", 4096, 4096, "equidistant", 360.0, 2047.5), src=("pano", 4096, 8192), rot=[], mask=0, pool=6, batch=1, pin="c2",
               text="c2: one 8192x4096 equirectangular frame -> 4096x4096 equidistant-360 inscribed per step"),
    "c3": dict(dst=("camera", 4096, 4096, "equisolid", 360.0, 2047.5), src=("camera", 4096, 4096, "equidistant", 360.0, 2047.5),
               rot=[(30.0, 45.0, 10.0)], mask=0, pool=8, batch=1, pin="c3",
               text="c3: one 4096x4096 equidistant-360 frame -> equisolid-360, rotation pitch 30 yaw 45 roll 10, per step"),
    "c5": dict(dst=("pano", 4096, 8192), src=("double", 3888, 7776, "equidistant", 180.0, None), rot=[], mask=2, pool=6, batch=1, pin="c5_180",
               text="c5: one 7776x3888 double-fisheye (2 x 180 degrees) frame -> 8192x4096 equirectangular stitch per step"),
}
CONFIGS["c4shard"] = dict(CONFIGS["c2"], pool=64, batch=8,
                          text="c4shard: one GPU's share of BASELINE config 4 - 64 distinct 8192x4096 panoramas resident, B frames per launch")
CONFIGS["c5shard"] = dict(CONFIGS["c5"], pool=32, batch=8,
                          text="c5shard: one GPU's share of BASELINE config 5 - 32 distinct 7776x3888 double-fisheye frames resident, B frames per launch")
# the LDS window budget each config is benchmarked (and profiled: profiles/traffic_<config>_<budget>.json) with: the
# library's default - the round-2 sweep found 7 KiB fastest or within noise of the fastest on every geometry
BENCH_BUDGET = {"c1": 7168, "c2": 7168, "c3": 7168, "c5": 7168, "c4shard": 7168, "c5shard": 7168}


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args) -> int:
    """Start args.gpus ranks of this script (one per GPU) and return their exit code.  Runs BEFORE any GPU call
    of this process: a process that has initialised HIP must not be replaced or forked into ranks."""
    import torch

    backend = os.environ.get("PB_DIST_BACKEND", "nccl")
    have = torch.cuda.device_count()  # counting devices does not initialise the runtime
    if backend == "nccl" and have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) visible (PB_DIST_BACKEND=gloo lets ranks share a GPU for rehearsal - at most 5 ranks on this pool: its boxes allow six GPU processes, this launcher included)", file=sys.stderr)
        return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def parse_args():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="c2")
    ap.add_argument("--batch", type=int, default=0, help="frames per launch (0 = the config's default)")
    ap.add_argument("--pool", type=int, default=0, help="distinct frames resident per GPU (0 = the config's default)")
    ap.add_argument("--budget", type=int, default=0, help="LDS window budget in bytes (0 = the value pinned for the config)")
    ap.add_argument("--tune", action="store_true", help="let the library pick the budget by timing (opt-in; reported in plan_create_ms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-events", action="store_true", help="skip per-launch HIP events (roofline from wall time)")
    ap.add_argument("--event-every", type=int, default=4, help="HIP event pairs bracket groups of this many consecutive timed launches")
    ap.add_argument("--sampling", choices=["nearest", "bilinear"], default="nearest",
                    help="nearest = the reference's truncating sampler (the headline); bilinear = the opt-in 4-tap mode (pb_remap_bilinear_u8; no reference behaviour)")
    ap.add_argument("--streams", type=int, default=1, help="HIP streams the (independent) launches are dealt to round-robin")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (the other BASELINE configs measured in the same process)")
    ap.add_argument("--no-live-traffic", action="store_true", help="roofline.traffic from the committed PMC pass (profiles/traffic_*.json) instead of two rocprofv3 --pmc child runs of this script")
    ap.add_argument("--flavour-probe", action="store_true", help="(child mode) print {plan_create_warm_ms, single_image_ms, faithful_kernel_ms} of --config for the math flavour this process loads (PB_MATH_FLAVOUR) and exit")
    ap.add_argument("--explain", action="store_true", help="also print the VERBOSE record (every figure with its workload text, timing protocol and notes) to stderr")
    ap.add_argument("--detail", default=None, help="write the verbose record to this file (default: gpurun_out/bench_detail.json when gpurun_out/ exists)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------
def build_projs(cfg):
    """pb_proj pair + rotation matrices of a config, through the host classes (so f_distance has the host's bits)."""
    import photonbend_amd as pb
    from photonbend_amd import _native as nat

    def one(p, role):
        kind, h, w = p[0], p[1], p[2]
        img = np.zeros((h, w, 3), np.uint8)
        if kind == "pano":
            return pb.PanoramaImage(img)._proj(role)
        lens = getattr(pb, p[3])()
        fov = pb.utils.to_radians(p[4])
        if kind == "camera":
            return pb.CameraImage(img, fov, lens, magnitude=p[5])._proj(role)
        return pb.DoubleCameraImage(img, fov, lens)._proj(role)

    rots = [pb.Rotation(*map(pb.utils.to_radians, r)).rotation_matrix for r in cfg["rot"]]
    return one(cfg["dst"], "dst"), rots, one(cfg["src"], "src")


def oracle_projs(cfg):
    from oracle import reference_path as orc  # CPU baseline leg only

    def one(p):
        if p[0] == "pano":
            return orc.Proj("pano", p[1], p[2])
        return orc.Proj(p[0], p[1], p[2], p[3], orc.to_radians(p[4]), p[5])

    rots = [tuple(map(orc.to_radians, r)) for r in cfg["rot"]]
    return one(cfg["dst"]), one(cfg["src"]), rots


def cpu_baseline(cfg, mpx_per_frame: float, min_seconds: float = 10.0, max_frames: int = 40):
    """The oracle (kind 'port') on this host: whole frames of the config until about `min_seconds` of CPU work
    have been timed (a bounded sample of the workload), then the gather alone with the coordinate map cached
    (what the GPU side amortises into its plan)."""
    from oracle import reference_path as orc
    from oracle.synth import synth_frame

    d, s, rots = oracle_projs(cfg)
    imgs = [synth_frame(s.height, s.width, frame=f, circle_mask=cfg["mask"]) for f in range(2)]
    frames = 0
    t0 = time.perf_counter()
    while True:
        out = orc.remap(d, s, imgs[frames % 2], rots)
        frames += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds or frames >= max_frames:
            break
    assert out.shape == (d.height, d.width, 3)
    res = {
        "value": round(frames * mpx_per_frame / dt, 3),
        "unit": "Mpx/s",
        "cores": 1,
        "kind": "port",
        "sample": f"{frames} full frame(s) of the config through oracle/reference_path.py (coordinate map recomputed per frame, as the reference does), {dt:.1f} s; host has {os.cpu_count()} logical cores, NumPy runs this path on 1",
    }
    # gather only: the reference's sampling stage with the coordinate map already there
    cmap = orc.coordinate_map(d)
    for r in rots:
        cmap = orc.rotate_map(orc.rotation_matrix(*r), cmap)
    n = 0
    t0 = time.perf_counter()
    while True:
        orc.sample(s, imgs[n % 2], cmap.copy() if s.kind == "pano" else cmap)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_seconds / 2 or n >= max_frames:
            break
    res["map_cached"] = {"value": round(n * mpx_per_frame / dt, 3), "unit": "Mpx/s", "cores": 1,
                         "sample": f"{n} frame(s), sampling stage only (process_coordinate_map on a precomputed float64 map), {dt:.1f} s"}
    return res


def _cpu_worker(job):
    cfg_name, frame_id = job
    from oracle import reference_path as orc
    from oracle.synth import synth_frame

    cfg = CONFIGS[cfg_name]
    d, s, rots = oracle_projs(cfg)
    img = synth_frame(s.height, s.width, frame=frame_id, circle_mask=cfg["mask"])
    t0 = time.perf_counter()
    orc.remap(d, s, img, rots)
    return time.perf_counter() - t0


def cpu_baseline_all_cores(cfg_name: str, mpx_per_frame: float, processes: int = 8):
    """One independent oracle process per core over distinct frames, capped at `processes` workers and skipped
    when the host lacks the memory (each worker peaks at several GB)."""
    try:
        import multiprocessing as mp

        import psutil

        n = max(1, min(processes, os.cpu_count() or 1))
        if psutil.virtual_memory().available < n * 8 * (1 << 30):
            return None
        ctx = mp.get_context("spawn")
        t0 = time.perf_counter()
        with ctx.Pool(n) as pool:
            pool.map(_cpu_worker, [(cfg_name, 100 + i) for i in range(n)])
        dt = time.perf_counter() - t0
        return {"value": round(n * mpx_per_frame / dt, 3), "unit": "Mpx/s", "cores": n,
                "sample": f"{n} processes x 1 full frame each, {dt:.1f} s wall including process start"}
    except Exception as exc:  # the all-cores figure is optional; never fail the bench over it
        return {"error": repr(exc)}


LINE = 128  # bytes the memory side moves per L2 miss (see byte_accounting)


def byte_accounting(plan, src_hw, device):
    """From the plan's own integer index map: algorithmic bytes per frame (3 B written per output pixel + 3 B read
    per in-bounds sample) and the bytes that MUST cross HBM: every 128-byte source line holding a sampled byte, once,
    + the output.  128 B is what the memory side moves per L2 miss on this chip whichever sectors are asked for
    (experiments/exp_calib.hip: one dword per 128-B line takes the time and the FETCH_SIZE of both its sectors, one dword
    per 256 B half of that), so a line is the unit a sample drags in."""
    import torch

    idx = plan.index_map(device=device).reshape(-1).long()
    valid = idx[idx >= 0]
    h, w = src_hw
    n_lines = (3 * h * w + LINE - 1) // LINE
    touched = torch.zeros(n_lines, dtype=torch.bool, device=device)
    touched[(3 * valid) // LINE] = True
    touched[(3 * valid + 2) // LINE] = True
    out_bytes = 3 * plan.dst.height * plan.dst.width
    return out_bytes + 3 * int(valid.numel()), out_bytes + LINE * int(touched.sum().item())


def measure_config(lib, nat, name, device, stream, steps=200, warmup=20, batch=0, pool_bytes=0, bilinear=False):
    """One config measured like the headline, inside this process: its plan at the pinned budget, a pool of distinct frames
    larger than the 256 MiB Infinity Cache, `warmup` untimed and `steps` timed launches (SURVEY 8d's protocol: >= 20 and >= 200) in
    groups of 4 between HIP event pairs on the launch stream.
    Returns the entry of the line's ``configs`` block."""
    import torch

    cfg = CONFIGS[name]
    batch = batch or cfg["batch"]
    d, rots, s = build_projs(cfg)
    budget = BENCH_BUDGET[name]
    plan_times, plan = [], None
    for _ in range(3):  # the first preparation of a geometry KIND in a process also pays its kernels' first launches; "warm" = the best of three
        plan = None  # (the previous plan's destruction - one device wait, its tables back to the library's block cache - stays outside the timed region)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        plan = nat.Plan(d, rots, s, budget=budget)
        torch.cuda.synchronize(device)
        plan_times.append((time.perf_counter() - t0) * 1e3)
    plan_ms = min(plan_times)
    sh, sw, dh, dw = s.height, s.width, d.height, d.width
    sbytes, dbytes = 3 * sh * sw, 3 * dh * dw
    pool = max(2 * batch, (pool_bytes or POOL_BYTES_MIN) // (sbytes + dbytes) + 1)
    pool = (pool + batch - 1) // batch * batch
    srcs = torch.empty((pool, sh, sw, 3), dtype=torch.uint8, device=device)
    for f in range(pool):
        nat.synth_frame(sh, sw, frame=1000 + f, seed=0, circle_mask=cfg["mask"], out=srcs[f])
    dsts = torch.empty((pool, dh, dw, 3), dtype=torch.uint8, device=device)
    sp0, dp0, h = srcs.data_ptr(), dsts.data_ptr(), plan.handle
    groups_in_pool = pool // batch

    fn = lib.pb_remap_bilinear_u8 if bilinear else lib.pb_remap_u8
    if bilinear:
        plan.ensure_bilinear()  # (plans are created for the reference's sampler; the opt-in mode's tables are built at its first use)

    def step(k):
        i = (k % groups_in_pool) * batch
        rc = fn(h, sp0 + i * sbytes, dp0 + i * dbytes, batch, sbytes, dbytes, stream)
        if rc:
            nat.check(rc)

    for k in range(warmup):
        step(k)
    every = 4
    n_groups = max(1, steps // every)
    ev = []
    for _ in range(2 * n_groups):
        e = ctypes.c_void_p()
        nat.check(lib.pb_event_create(ctypes.byref(e)))
        ev.append(e)
    torch.cuda.synchronize(device)
    for g in range(n_groups):
        lib.pb_event_record(ev[2 * g], stream)
        for k in range(every):
            step(g * every + k)
        lib.pb_event_record(ev[2 * g + 1], stream)
    torch.cuda.synchronize(device)
    ms = ctypes.c_float()
    durs = []
    for g in range(n_groups):
        nat.check(lib.pb_event_elapsed_ms(ev[2 * g], ev[2 * g + 1], ctypes.byref(ms)))
        durs.append(ms.value / every / batch)
    for e in ev:
        lib.pb_event_destroy(e)
    if bilinear:  # the opt-in 4-tap mode (no reference behaviour; its parity is pinned to OUR definition: tests/test_hip_bilinear.py)
        per_frame_ms = float(np.mean(durs))
        alg, must = byte_accounting(plan, (sh, sw), device)
        alg4 = alg + 3 * ((alg - 3 * dh * dw) // 3) * 3  # four taps of 3 bytes per in-bounds sample instead of one
        info = plan.info()
        out = {"workload": cfg["text"] + " - OPT-IN bilinear sampling (pb_remap_bilinear_u8; not the reference's sampler)", "frames_per_launch": batch,
               "kernel_ms_per_frame": round(per_frame_ms, 5), "kernel_ms_per_frame_p10": round(float(np.percentile(durs, 10)), 5),
               "kernel_ms_per_frame_p90": round(float(np.percentile(durs, 90)), 5),
               "mpx_per_s": round(dh * dw / 1e6 / (per_frame_ms * 1e-3), 1), "launches_timed": n_groups * every,
               "algorithmic_bytes_per_frame": alg4, "must_move_bytes_per_frame": must,
               "frac_must_move": round(must / (per_frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "frac_4tap": round(alg4 / (per_frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "frac_note": "frac_must_move = the bytes that MUST cross HBM (every 128-byte source line holding a sample, once, + the output: the nearest mode's) / kernel time / 8 TB/s - the roofline figure; "
                            "frac_4tap prices four taps per sample (3 B written per output pixel + 4 x 3 B read per in-bounds sample): arithmetic riding on that traffic, not a roofline fraction",
               "traffic_bytes_per_frame": traffic_for(name + "_bilinear", info)[0],
               "bilinear_float64_tiles": info["bilinear_float64_tiles"], "tiles": info["tiles"], "tile_mix": plan.bilinear_tile_mix(), "launch_shape": plan.bilinear_launch_shape(),
               "nearest_over_bilinear_note": "one launch per call: tile models where certified to 1/1024 px, the plan's exact coordinate table elsewhere; no float64 per frame"}
        del srcs, dsts, plan
        torch.cuda.empty_cache()
        return out
    alg, must = byte_accounting(plan, (sh, sw), device)
    pins = json.load(open(os.path.join(ROOT, "tests", "golden", "full.json")))
    ref_alg = int(pins[cfg["pin"]]["algorithmic_bytes"])
    if alg != ref_alg:
        raise SystemExit(f"bench.py: {name}: algorithmic bytes from the plan's index map ({alg}) differ from the reference's ({ref_alg})")
    per_frame_ms = float(np.mean(durs))
    info = plan.info()
    wall = {}
    if batch == 1:
        for n in (1, 2, 3):
            wall[str(n)] = round(multi_stream_ms(fn, h, sp0, dp0, sbytes, dbytes, pool, batch, device, n), 5)
    traffic, _ = traffic_for(name, info)
    if traffic and batch > 1:
        traffic = traffic / batch
    out = {
        "workload": cfg["text"],
        "frames_per_launch": batch,
        "frames_resident": pool,
        "kernel_ms_per_frame": round(per_frame_ms, 5),
        "kernel_ms_per_frame_p10": round(float(np.percentile(durs, 10)), 5),
        "kernel_ms_per_frame_p90": round(float(np.percentile(durs, 90)), 5),
        "mpx_per_s": round(dh * dw / 1e6 / (per_frame_ms * 1e-3), 1),
        "algorithmic_bytes_per_frame": alg,
        "must_move_bytes_per_frame": must,
        "frac": round(alg / (per_frame_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
        "attainable_frac": round(alg / must, 4),
        "traffic_bytes_per_frame": traffic,
        "plan_create_warm_ms": round(plan_ms, 3),
        "plan_create_first_ms": round(plan_times[0], 3),
        "tiles": {k: info[k] for k in ("tiles", "lean_tiles", "direct_tiles", "black_tiles", "fix_tiles")},
        "window_budget": info["window_budget"],
        "launches_timed": n_groups * every,
    }
    if wall:
        out["wall_ms_per_frame_by_streams"] = wall
    del srcs, dsts, plan
    torch.cuda.empty_cache()
    return out


TILE_MIX_KEYS = ("tiles", "fix_tiles", "fix_pixels", "lean_tiles", "black_tiles", "direct_tiles", "window_budget")


def traffic_for(name, info):
    """HBM bytes per launch from the committed PMC pass (profiles/traffic_<config>_<budget>.json: FETCH_SIZE x 2 + WRITE_SIZE, separate
    --pmc runs) - or (None, why) when that file describes ANOTHER tile mix than the running plan's (a counter figure
    taken on a plan with other tile classes says nothing about this launch)."""
    tpath = os.path.join(ROOT, "profiles", f"traffic_{name}_{info['window_budget']}.json")
    if not os.path.exists(tpath):
        return None, "no PMC pass committed for this config and budget"
    try:
        tj = json.load(open(tpath))
    except Exception as exc:
        return None, repr(exc)
    theirs = tj.get("plan") or {}
    diff = [k for k in TILE_MIX_KEYS if theirs.get(k) != info.get(k)]
    if diff:
        return None, f"{os.path.relpath(tpath, ROOT)} was taken on another tile mix ({', '.join(f'{k}: {theirs.get(k)} vs {info.get(k)}' for k in diff)})"
    return tj.get("hbm_bytes_per_launch"), os.path.relpath(tpath, ROOT)


def live_traffic(args, batch, kernel_prefix, timeout_s=180):
    """HBM bytes per launch of the dominant kernel, MEASURED for this run: FETCH_SIZE and WRITE_SIZE, each in its own
    `rocprofv3 --kernel-trace --pmc <counter>` pass of a short child run of this script (same config, sampling, budget, batch; 12 launches),
    corrected as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE x 2, WRITE_SIZE as reported; both in KiB).  -> (bytes or None, source)"""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    got = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="pb_pmc_", dir="/tmp")
        cmd = [exe, "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d, "--", sys.executable, os.path.abspath(__file__), "--config", args.config,
               "--sampling", args.sampling, "--batch", str(batch), "--steps", "12", "--warmup", "2", "--no-cpu-baseline", "--no-configs", "--no-events", "--no-live-traffic"]
        if args.budget:
            cmd += ["--budget", str(args.budget)]
        try:
            res = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row["Counter_Name"] == ctr and row["Kernel_Name"].split("(")[0].replace("void ", "").startswith(kernel_prefix):
                        vals.append(float(row["Counter_Value"]))
            if res.returncode != 0 or not vals:
                return None, f"rocprofv3 --pmc {ctr}: rc {res.returncode}, {len(vals)} dispatches of {kernel_prefix}: {res.stderr[-200:]!r}"
            got[ctr] = (sum(vals) / len(vals), len(vals))
        except Exception as exc:  # a measurement extra: the committed pass stands in
            return None, f"rocprofv3 --pmc {ctr}: {exc!r}"
        finally:
            shutil.rmtree(d, ignore_errors=True)
    hbm = int(2 * got["FETCH_SIZE"][0] * 1024 + got["WRITE_SIZE"][0] * 1024)
    return hbm, f"live: rocprofv3 --pmc FETCH_SIZE ({got['FETCH_SIZE'][1]} dispatches) x 2 + WRITE_SIZE ({got['WRITE_SIZE'][1]}), separate passes of this command with --steps 12"


def headline_extras(lib, nat, d, rots, s, cfg, device, stream, budget):
    """single_image_ms: what ONE image of a new geometry costs (warm plan creation + its first frame, device-resident
    input); faithful_kernel_ms: the float64 chain of the same geometry (what a deferred plan runs); and what the opt-in bilinear
    mode's tables add to a plan at that mode's first use (ms)."""
    import torch

    src = nat.synth_frame(s.height, s.width, frame=77, seed=0, circle_mask=cfg["mask"])
    out = torch.empty((d.height, d.width, 3), dtype=torch.uint8, device=device)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    plan = nat.Plan(d, rots, s, budget=budget)
    nat.check(lib.pb_remap_u8(plan.handle, src.data_ptr(), out.data_ptr(), 1, 0, 0, stream))
    torch.cuda.synchronize(device)
    single_ms = (time.perf_counter() - t0) * 1e3
    ts = []
    for k in range(3):  # what the opt-in mode adds to a plan at its first use (the best of three plans: the first also loads the mode's plan kernels)
        p2 = plan if k == 0 else nat.Plan(d, rots, s, budget=budget)
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        p2.ensure_bilinear()
        torch.cuda.synchronize(device)
        ts.append((time.perf_counter() - t0) * 1e3)
    bilinear_prepare_ms = round(min(ts), 3)
    plan.set_mode(nat.MODE_FAITHFUL)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    nat.check(lib.pb_event_create(ctypes.byref(e0)))
    nat.check(lib.pb_event_create(ctypes.byref(e1)))
    for _ in range(2):
        nat.check(lib.pb_remap_u8(plan.handle, src.data_ptr(), out.data_ptr(), 1, 0, 0, stream))
    lib.pb_event_record(e0, stream)
    for _ in range(8):
        nat.check(lib.pb_remap_u8(plan.handle, src.data_ptr(), out.data_ptr(), 1, 0, 0, stream))
    lib.pb_event_record(e1, stream)
    nat.check(lib.pb_event_sync(e1))
    ms = ctypes.c_float()
    nat.check(lib.pb_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
    lib.pb_event_destroy(e0)
    lib.pb_event_destroy(e1)
    return round(single_ms, 3), round(ms.value / 8, 5), bilinear_prepare_ms



def multi_stream_ms(fn, handle, sp0, dp0, sbytes, dbytes, n_pool, batch, device, n_streams, launches=120):
    """Wall-clock ms per frame of `launches` independent launches dealt round-robin to n_streams HIP streams (device drained
    before and after): the next launch's ramp runs in the previous one's drain.  A per-kernel duration is not defined there -
    the kernels overlap - so this is reported NEXT to the single-stream figures the roofline block is computed from."""
    import torch

    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    sts = [int(x.cuda_stream) for x in streams]
    groups = n_pool // batch

    def go(n):
        for k in range(n):
            i = (k % groups) * batch
            rc = fn(handle, sp0 + i * sbytes, dp0 + i * dbytes, batch, sbytes, dbytes, sts[k % n_streams])
            if rc:
                raise RuntimeError(f"launch failed: {rc}")

    go(2 * n_streams + 8)
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    go(launches)
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) * 1e3 / (launches * batch)


def scattered_batch(lib, nat, plan, cfg, s, d, device, stream, n=8, launches=24):
    """a ring of SEPARATELY ALLOCATED frames (no common stride) through pb_remap_u8v - ONE launch per `n` frames on one
    stream.  2 x n frame pairs (beyond the Infinity Cache for the headline config) with decoy allocations in between; kernel ms per frame
    between HIP events, next to the same frames as n single pb_remap_u8 launches."""
    import torch

    pairs, decoys = [], []
    for k in range(2 * n):
        pairs.append((nat.synth_frame(s.height, s.width, frame=3000 + k, seed=0, circle_mask=cfg["mask"]),
                      torch.empty((d.height, d.width, 3), dtype=torch.uint8, device=device)))
        decoys.append(torch.empty(4096 * (1 + k % 3) + 16, dtype=torch.uint8, device=device))
    tabs = []
    for a in (0, n):
        sp = (ctypes.c_void_p * n)(*[int(p[0].data_ptr()) for p in pairs[a : a + n]])
        dp = (ctypes.c_void_p * n)(*[int(p[1].data_ptr()) for p in pairs[a : a + n]])
        tabs.append((sp, dp))
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    nat.check(lib.pb_event_create(ctypes.byref(e0)))
    nat.check(lib.pb_event_create(ctypes.byref(e1)))
    ms = ctypes.c_float()

    def timed(fn):
        for k in range(4):
            fn(k)
        lib.pb_event_record(e0, stream)
        for k in range(launches):
            fn(k)
        lib.pb_event_record(e1, stream)
        nat.check(lib.pb_event_sync(e1))
        nat.check(lib.pb_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        return ms.value / (launches * n)

    def vec(k):
        sp, dp = tabs[k & 1]
        nat.check(lib.pb_remap_u8v(plan.handle, sp, dp, n, stream))

    def singles(k):
        for p in pairs[(k & 1) * n : (k & 1) * n + n]:
            nat.check(lib.pb_remap_u8(plan.handle, p[0].data_ptr(), p[1].data_ptr(), 1, 0, 0, stream))

    out = {"frames_per_launch": n, "separately_allocated_frames": 2 * n, "u8v_ms_per_frame": round(timed(vec), 5),
           "single_launches_ms_per_frame": round(timed(singles), 5),
           "note": "pb_remap_u8v: frame pointers in the kernel-argument segment, one launch per 8 frames on one stream; byte-identical to the single launches (tests/test_hip_plan.py)"}
    lib.pb_event_destroy(e0)
    lib.pb_event_destroy(e1)
    del pairs, decoys
    torch.cuda.empty_cache()
    return out


def graph_replay_ms(lib, nat, plan, srcs, dsts, sbytes, dbytes, n_pool, device, launches=8, replays=25):
    """`launches` single-frame launches of the headline config captured into ONE hipGraph and replayed -
    ms per frame over `replays` replays (distinct frames of the pool inside a graph; the same graph replayed)."""
    import torch

    side = torch.cuda.Stream(device=device)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            st = int(torch.cuda.current_stream(device).cuda_stream)
            for k in range(launches):
                i = k % n_pool
                nat.check(lib.pb_remap_u8(plan.handle, srcs.data_ptr() + i * sbytes, dsts.data_ptr() + i * dbytes, 1, sbytes, dbytes, st))
        for _ in range(3):
            g.replay()
        side.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(side)
        for _ in range(replays):
            g.replay()
        e1.record(side)
        side.synchronize()
    return e0.elapsed_time(e1) / (replays * launches)


def host_path(cfg, d, rots, s):
    """The drop-in's REAL path for the headline config: NumPy in -> NumPy out through the facade.  The
    reference's contract is ndarray in, fresh ndarray out (core/__init__.py:66-92), so a user who swaps imports pays upload +
    kernel + download per frame - never `value`, always next to it."""
    import photonbend_amd as pb
    from photonbend_amd import _device, _hostpipe, batch
    from photonbend_amd import _native as nat

    sh, dh = (s.height, s.width, 3), (d.height, d.width, 3)
    n_pool = 4
    rng = np.random.default_rng(7)
    pool = [rng.integers(0, 256, size=sh, dtype=np.uint8) for _ in range(n_pool)]
    plan = nat.Plan(d, rots, s)

    def med_ms(ts):
        return round(sorted(ts)[len(ts) // 2] * 1e3, 3)

    def singles(make, reps=8):
        ts = []
        for k in range(reps):
            a = make(k)
            t0 = time.perf_counter()
            out = _hostpipe.remap_ndarray(plan, a)
            ts.append(time.perf_counter() - t0)
        assert out.shape == dh
        return med_ms(ts[2:])

    fresh = singles(lambda k: pool[k % n_pool].copy())  # (the copy is made outside the timed region)
    buf = np.empty(sh, np.uint8)

    def refill(k):
        buf[...] = pool[k % n_pool]
        return buf

    reused = singles(refill)
    list(batch.remap_frames(plan, (pool[k % n_pool] for k in range(2 * n_pool))))  # (ring buffers: page-locked after their second sighting)
    n = 16
    t0 = time.perf_counter()
    cnt = sum(1 for _ in batch.remap_frames(plan, (pool[k % n_pool] for k in range(n))))
    ring = (time.perf_counter() - t0) / n * 1e3
    once = [pool[k % n_pool].copy() for k in range(n)]
    t0 = time.perf_counter()
    cnt += sum(1 for _ in batch.remap_frames(plan, once))  # (a list: frames that already exist are page-locked one ahead of their upload)
    staged = (time.perf_counter() - t0) / n * 1e3
    assert cnt == 2 * n
    del once
    lib = nat.load()
    pipe = _hostpipe.pipe_for()
    hin = _device.PINNED.ndarray(sh, np.uint8)
    hout = _device.PINNED.ndarray(dh, np.uint8)
    din, dout = _device.DeviceArray((hin.nbytes,), np.uint8), _device.DeviceArray((hout.nbytes,), np.uint8)
    dma = {}
    for key, fn in (("h2d_ms", lambda: lib.pb_memcpy_h2d(din.data_ptr(), hin.ctypes.data, hin.nbytes, pipe.stream.handle)),
                    ("d2h_ms", lambda: lib.pb_memcpy_d2h(hout.ctypes.data, dout.data_ptr(), hout.nbytes, pipe.stream.handle))):
        fn()
        pipe.stream.sync()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        pipe.stream.sync()
        dma[key] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    return {
        "workload": cfg["text"] + " - uint8 ndarray in host memory in, fresh uint8 ndarray out",
        "ms_per_frame_single_call": fresh,
        "ms_per_frame_single_call_note": "remap of an ndarray never seen before (frame-sized: page-locked in place, 0.2-0.35 ms, then ONE upload DMA) -> fresh ndarray (page-locked, written by the download DMA directly)",
        "ms_per_frame_single_call_reused_buffer": reused,
        "ms_per_frame_single_call_reused_buffer_note": "the caller refills ONE buffer: page-locked in place when first seen, then one DMA straight out of the caller's memory per call",
        "ms_per_frame_streamed": round(max(ring, staged), 3),
        "ms_per_frame_streamed_ring_of_caller_buffers": round(ring, 3),
        "ms_per_frame_streamed_never_seen_arrays": round(staged, 3),
        "ms_per_frame_streamed_note": "batch.remap_frames: the upload DMA of frame k + 1 beside the remap kernel of frame k, which stores over PCIe straight into its page-locked result ndarray (the two DMA directions side by side take 2.41 ms on this pool's boxes, an upload DMA beside a storing kernel 2.06: experiments/r6/pcie_paths.py); `ms_per_frame_streamed` is the slower of the two source kinds",
        **dma,
        "bytes_up": int(np.prod(sh)), "bytes_down": int(np.prod(dh)),
        "mpx_per_s_streamed": round(d.height * d.width / 1e6 / (max(ring, staged) * 1e-3), 1),
        "torch_in_the_path": False,
    }


def sharded_workload(lib, nat, parallel, name, per_rank, rank, world, device, coll_device, stream, dist, passes=3, chunk=8):
    """BASELINE configs 4 / 5 as north_star states them: a batch of per_rank x world distinct frames sharded contiguously over the
    ranks (parallel.shard_range), every rank remapping ITS frames, resident in its own HBM, `chunk` per launch; the only collective
    on the data path is the broadcast of the parameter block.  Returns this rank's view: (seconds for `passes` passes over its
    share - max-reduced by the caller -, frames per pass, SHA-256 of its first output frame, global index of that frame, plan)."""
    import hashlib

    import torch

    cfg = CONFIGS[name]
    block = None
    if rank == 0:
        d, rots, s = build_projs(cfg)
        block = parallel.pack_params(d, rots, s)
    d, rots, s = parallel.unpack_params(parallel.broadcast_params(block, device=coll_device, src=0))
    plan = nat.Plan(d, rots, s, budget=BENCH_BUDGET[name])
    total = per_rank * world
    mine = parallel.shard_range(total, world, rank)
    n = len(mine)
    sh, sw, dh, dw = s.height, s.width, d.height, d.width
    sbytes, dbytes = 3 * sh * sw, 3 * dh * dw
    srcs = torch.empty((n, sh, sw, 3), dtype=torch.uint8, device=device)
    for i, k in enumerate(mine):
        nat.synth_frame(sh, sw, frame=k, seed=0, circle_mask=cfg["mask"], out=srcs[i])
    dsts = torch.empty((n, dh, dw, 3), dtype=torch.uint8, device=device)

    def one_pass():
        for a in range(0, n, chunk):
            m = min(chunk, n - a)
            rc = lib.pb_remap_u8(plan.handle, srcs.data_ptr() + a * sbytes, dsts.data_ptr() + a * dbytes, m, sbytes, dbytes, stream)
            if rc:
                nat.check(rc)

    one_pass()
    torch.cuda.synchronize(device)
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(passes):
        one_pass()
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    digest = hashlib.sha256(dsts[0].cpu().numpy().tobytes()).digest()
    del srcs, dsts
    torch.cuda.empty_cache()
    return dt, n, digest, mine.start, plan, (d, s, cfg)


LINE_LIMIT = 6144  # the driver keeps the parsed core of the line plus a few KB of stdout tail: every figure must fit
CONFIGS_KEYS = ["kernel_us_per_frame", "frac_of_8TBs_algorithmic_bytes (bilinear rows: 4-tap bytes)", "frac_of_8TBs_must_move_bytes", "plan_create_warm_ms (bilinear rows: null)"]


def _us(ms):
    return None if ms is None else round(ms * 1e3, 2)


def compact_line(full):
    """The ONE line rank 0 prints: every measured figure, numbers only, under LINE_LIMIT bytes.  The verbose record (workload texts, timing
    protocol, notes, tile mixes, plan info) goes to --detail / --explain; DESIGN.md section 4 says what each key means."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    out = {k: full[k] for k in keep}
    c = full["config"]
    out["config"] = {"workload": c["workload"], "name": c["name"], "frames_per_launch": c["frames_per_launch"], "frames_resident_per_gpu": c["frames_resident_per_gpu"],
                     "streams": c["streams"], "sampling": c["sampling"].split(",")[0].split(" ")[0], "parallelism": f"dp{full['n_gpus']}"}
    r = full["roofline"]
    out["roofline"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "copy_ceiling_gbs", "frac_of_copy_ceiling", "algorithmic_bytes_per_launch",
                                         "must_move_bytes_per_launch", "attainable_frac", "kernel_ms_mean", "kernel_ms_median", "kernel_ms_p10", "kernel_ms_p90", "window_budget")}
    out["roofline"]["traffic_live"] = r.get("traffic_live", False)
    if "frac_unamortised" in r:
        out["roofline"]["frac_unamortised"] = {k: r["frac_unamortised"][k] for k in ("single_image", "faithful_kernel", "break_even_frames")}
    for k in ("plan_create_ms", "plan_create_warm_ms", "plan_bilinear_prepare_ms", "first_frame_ms", "single_image_ms", "faithful_kernel_ms"):
        if k in full:
            out[k] = full[k]
    if "configs" in full:  # all ten entries BEFORE the long optional blocks
        rows = {}
        for name, e in full["configs"].items():
            rows[name] = [_us(e["kernel_ms_per_frame"]), e.get("frac", e.get("frac_4tap")), e.get("frac_must_move", round(e["must_move_bytes_per_frame"] / (e["kernel_ms_per_frame"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)),
                          e.get("plan_create_warm_ms")]
        out["configs_keys"] = CONFIGS_KEYS
        out["configs"] = rows
    if "flavours" in full:
        out["flavours"] = full["flavours"]
    sb = full.get("scattered_batch")
    if sb:
        out["scattered_batch"] = sb if "error" in sb else {"frames_per_launch": sb["frames_per_launch"], "u8v_us_per_frame": _us(sb["u8v_ms_per_frame"]),
                                                           "single_launches_us_per_frame": _us(sb["single_launches_ms_per_frame"])}
    if "graph_replay_ms_per_frame" in full:
        out["graph_replay_us_per_frame"] = _us(full["graph_replay_ms_per_frame"])
    w = full.get("wall_ms_per_frame_by_streams")
    if w:
        out["wall_us_per_frame_by_streams"] = {k: _us(v) for k, v in w.items()}
    hp = full.get("host_path")
    if hp:
        out["host_path"] = hp if "error" in hp else {k: v for k, v in hp.items() if not k.endswith("_note") and k not in ("workload", "torch_in_the_path")}
    sh = full.get("sharded")
    if sh:
        out["sharded"] = {k: ({kk: v[kk] for kk in ("frames_total", "frames_per_gpu", "frames_per_launch", "mpx_per_s", "ms_per_frame_per_gpu", "first_frames_identical_to_rank0")}
                              if isinstance(v, dict) else v) for k, v in sh.items() if k != "collective"}
    cb = full.get("cpu_baseline")
    if cb:
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"].split(" (")[0] + f", host {os.cpu_count()} cores"}
        if "map_cached" in cb:
            out["cpu_baseline"]["map_cached_mpx_per_s"] = cb["map_cached"]["value"]
        ac = cb.get("all_cores")
        if ac and "value" in ac:
            out["cpu_baseline"]["all_cores"] = {"value": ac["value"], "cores": ac["cores"]}
    out["ranks_seen"] = full.get("ranks_seen")
    out["collective_backend"] = full.get("collective_backend")
    return out


def copy_ceiling_gbs(lib, nat, device, stream) -> float:
    """A plain 16-byte-per-lane device copy of 512 MiB (beyond the 256 MiB Infinity Cache), read + write bytes per second."""
    import torch

    n = 512 << 20
    a = torch.empty(n, dtype=torch.uint8, device=device)
    b = torch.empty(n, dtype=torch.uint8, device=device)
    a.random_(0, 255)
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    nat.check(lib.pb_event_create(ctypes.byref(e0)))
    nat.check(lib.pb_event_create(ctypes.byref(e1)))
    for _ in range(3):
        nat.check(lib.pb_stream_copy(b.data_ptr(), a.data_ptr(), n, stream))
    best = 1e30
    ms = ctypes.c_float()
    for _ in range(5):
        lib.pb_event_record(e0, stream)
        nat.check(lib.pb_stream_copy(b.data_ptr(), a.data_ptr(), n, stream))
        lib.pb_event_record(e1, stream)
        nat.check(lib.pb_event_sync(e1))
        nat.check(lib.pb_event_elapsed_ms(e0, e1, ctypes.byref(ms)))
        best = min(best, ms.value)
    lib.pb_event_destroy(e0)
    lib.pb_event_destroy(e1)
    del a, b
    return 2 * n / (best * 1e-3) / 1e9


def flavour_probe(args):
    """Child mode: what ONE math flavour costs on a geometry - warm plan preparation (thresholds, models, certification against
    this flavour's float64 chain, tables) and the float64 kernel per frame.  The flavour is the library the process loads (PB_MATH_FLAVOUR)."""
    import torch

    from photonbend_amd import _native as nat

    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    lib = nat.load()
    cfg = CONFIGS[args.config]
    d, rots, s = build_projs(cfg)
    times = []
    for _ in range(4):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        plan = nat.Plan(d, rots, s, budget=BENCH_BUDGET[args.config])
        torch.cuda.synchronize(device)
        times.append((time.perf_counter() - t0) * 1e3)
        del plan
    single_ms, faithful_ms, _ = headline_extras(lib, nat, d, rots, s, cfg, device, 0, BENCH_BUDGET[args.config])
    print(json.dumps({"flavour": nat.MATH_FLAVOUR, "lib": os.path.basename(nat.LIB_PATH), "config": args.config, "plan_create_warm_ms": round(min(times[1:]), 3),
                      "single_image_ms": single_ms, "faithful_kernel_ms": faithful_ms}), flush=True)


def flavours_block(config="c3", timeout_s=150):
    """plan_create_warm_ms / faithful_kernel_ms of `config` under BOTH math flavours: two child runs of this script (a process loads one library)."""
    out = {"config": config, "keys": ["plan_create_warm_ms", "single_image_ms", "faithful_kernel_ms"]}
    for fl in ("svml", "libm"):
        env = dict(os.environ, PB_MATH_FLAVOUR=fl)
        env.pop("PB_LIB_PATH", None)
        try:
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "--flavour-probe", "--config", config], env=env, capture_output=True, text=True, timeout=timeout_s)
            j = json.loads(res.stdout.strip().splitlines()[-1])
            assert j["flavour"] == fl
            out[fl] = [j["plan_create_warm_ms"], j["single_image_ms"], j["faithful_kernel_ms"]]
        except Exception as exc:  # a measurement extra: never fail the line over it
            out[fl] = repr(exc)[:120]
    return out


def main():
    args = parse_args()
    if args.flavour_probe:
        return flavour_probe(args)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args))

    import torch
    import torch.distributed as dist

    from photonbend_amd import _native as nat
    from photonbend_amd import parallel

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("PB_BENCH_FAIL_RANK") == str(rank) and world > 1:
        # rehearsal aid (tests/test_bench_launcher.py): this rank dies before it joins the group - the launcher must end the other ranks
        # and the whole command must exit non-zero, never print a line for fewer ranks than asked for
        print(f"bench.py: rank {rank} exits on request (PB_BENCH_FAIL_RANK)", file=sys.stderr)
        os._exit(3)
    cfg = CONFIGS[args.config]
    batch = args.batch or cfg["batch"]
    pool = args.pool or max(cfg["pool"], POOL_BYTES_MIN // (3 * (cfg["src"][1] * cfg["src"][2] + cfg["dst"][1] * cfg["dst"][2])) + 1)
    pool = max(batch, (pool // batch) * batch)  # launches take `batch` consecutive frames of the pool
    budget = args.budget or BENCH_BUDGET[args.config]

    # one rank per GPU; PB_DIST_BACKEND=gloo + fewer GPUs than ranks is only for rehearsing the multi-rank code
    # path on a one-GPU box (ranks then share a device and the collectives run over gloo on CPU tensors)
    backend = os.environ.get("PB_DIST_BACKEND", "nccl")
    local = local % max(1, torch.cuda.device_count()) if backend != "nccl" else local
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    from photonbend_amd.utils import numa

    cpus_as_found = os.sched_getaffinity(0) if hasattr(os, "sched_getaffinity") else None
    numa_cpus = 0 if os.environ.get("PB_BENCH_NO_NUMA_PIN") == "1" else numa.pin_to_device(local)  # (the rank runs - and first-touches its host buffers - on its GPU's NUMA node: host_path's upload DMA)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    force_dist = os.environ.get("PB_FORCE_DIST") == "1"  # a group of ONE rank: the RCCL broadcast really runs on a 1-GPU box
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        kw = {} if "RANK" in os.environ else {"rank": 0, "world_size": 1}
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device, **kw)  # RCCL over xGMI
        else:
            dist.init_process_group(backend=backend, **kw)

    # rank 0 owns the parameters; everyone else receives the block over RCCL
    block = None
    if rank == 0:
        d, rots, s = build_projs(cfg)
        block = parallel.pack_params(d, rots, s)
    block = parallel.broadcast_params(block, device=coll_device, src=0)
    d, rots, s = parallel.unpack_params(block)
    lib = nat.load()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    plan = nat.Plan(d, rots, s, tune=args.tune, budget=0 if args.tune else budget)
    torch.cuda.synchronize(device)
    plan_create_ms = (time.perf_counter() - t0) * 1e3  # cold: the first plan of the process also loads the code object
    warm_times = []
    for _ in range(3):  # what every further geometry costs: thresholds, models, certification, tables - best of three like the other configs'
        t0 = time.perf_counter()  # (each destroyed before the next: from the second on its tables come out of the library's block cache)
        warm = nat.Plan(d, rots, s, budget=budget)
        torch.cuda.synchronize(device)
        warm_times.append((time.perf_counter() - t0) * 1e3)
        del warm
        torch.cuda.synchronize(device)
    plan_create_warm_ms = min(warm_times)
    sh, sw, dh, dw = s.height, s.width, d.height, d.width
    mpx_per_frame = dh * dw / 1e6

    # per-rank frame pool, generated on the device (frame ids disjoint across ranks)
    srcs = torch.empty((pool, sh, sw, 3), dtype=torch.uint8, device=device)
    for f in range(pool):
        nat.synth_frame(sh, sw, frame=rank * pool + f, seed=0, circle_mask=cfg["mask"], out=srcs[f])
    dsts = torch.empty((pool, dh, dw, 3), dtype=torch.uint8, device=device)
    n_streams = max(1, args.streams)
    streams = [torch.cuda.Stream(device=device) for _ in range(n_streams)]
    sts = [int(x.cuda_stream) for x in streams]
    sbytes, dbytes = 3 * sh * sw, 3 * dh * dw
    sp0, dp0 = srcs.data_ptr(), dsts.data_ptr()
    h = plan.handle
    groups_in_pool = pool // batch

    bilinear = args.sampling == "bilinear"
    remap_fn = lib.pb_remap_bilinear_u8 if bilinear else lib.pb_remap_u8
    if bilinear:
        plan.ensure_bilinear()

    def step(k):
        i = (k % groups_in_pool) * batch
        rc = remap_fn(h, sp0 + i * sbytes, dp0 + i * dbytes, batch, sbytes, dbytes, sts[k % n_streams])
        if rc:
            nat.check(rc)

    def sync_all():
        torch.cuda.synchronize(device)
        if dist.is_initialized():
            dist.barrier()
        torch.cuda.synchronize(device)

    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    step(0)
    torch.cuda.synchronize(device)
    first_frame_ms = (time.perf_counter() - t0) * 1e3
    for k in range(args.warmup):
        step(k)
    K = args.steps
    use_events = not args.no_events
    ev = []
    # an event pair costs a few microseconds on the stream, so one pair brackets a GROUP of consecutive
    # launches (same stream, back to back): the kernel duration is still measured live inside the timed
    # region, group time / group size, and the wall clock (value) is not inflated by per-launch events
    every = max(1, min(args.event_every, K))
    groups = [(g, min(g + every, K)) for g in range(0, K, every)] if use_events else []
    for _ in range(2 * len(groups)):
        e = ctypes.c_void_p()
        nat.check(lib.pb_event_create(ctypes.byref(e)))
        ev.append(e)
    sync_all()
    t0 = time.perf_counter()
    if use_events:
        for n, (a, b) in enumerate(groups):
            stn = sts[0] if n_streams == 1 else sts[a % n_streams]
            lib.pb_event_record(ev[2 * n], stn)
            for k in range(a, b):
                step(k)
            lib.pb_event_record(ev[2 * n + 1], stn)
    else:
        for k in range(K):
            step(k)
    # closing bracket: this rank's clock stops when ITS device has drained; the barrier follows and the
    # MAX over ranks is what gets reported, so no rank's time hides behind another's barrier latency
    torch.cuda.synchronize(device)
    dt = time.perf_counter() - t0
    sync_all()

    t = torch.tensor([dt], dtype=torch.float64, device=coll_device)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt_max = float(t.item())

    durs = []
    if use_events:
        ms = ctypes.c_float()
        for n, (a, b) in enumerate(groups):
            nat.check(lib.pb_event_elapsed_ms(ev[2 * n], ev[2 * n + 1], ctypes.byref(ms)))
            durs.append(ms.value / (b - a))
        for e in ev:
            lib.pb_event_destroy(e)

    if rank == 0:
        frames_total = world * K * batch
        value = frames_total * mpx_per_frame / dt_max
        launch_ms = float(np.mean(durs)) if durs else dt_max / K * 1e3
        alg_frame, must_frame = byte_accounting(plan, (sh, sw), device)
        pins = json.load(open(os.path.join(ROOT, "tests", "golden", "full.json")))
        ref_alg = int(pins[cfg["pin"]]["algorithmic_bytes"])
        if alg_frame != ref_alg:
            raise SystemExit(f"bench.py: algorithmic bytes from the plan's index map ({alg_frame}) differ from the reference's ({ref_alg})")
        if bilinear:  # four taps of 3 bytes per in-bounds sample instead of one (must-move: the same lines, give or take a rim)
            alg_frame += 3 * ((alg_frame - 3 * dh * dw) // 3) * 3
        alg_launch = alg_frame * batch
        achieved = alg_launch / (launch_ms * 1e-3) / 1e9
        ceiling = copy_ceiling_gbs(lib, nat, device, sts[0])
        info = plan.info()
        traffic, traffic_src = (None, "bilinear run: the committed PMC passes are the nearest kernel's") if bilinear else traffic_for(args.config, info)
        if world == 1 and not args.no_live_traffic:
            kern = ("pb_bilinear_double_hot_kernel" if cfg["src"][0] == "double" else "pb_bilinear_hot_kernel") if bilinear else ("pb_hot_double_kernel" if cfg["src"][0] == "double" else "pb_hot_win_kernel")
            live, live_src = live_traffic(args, batch, kern)
            if live is not None:
                committed = traffic
                traffic, traffic_src = live, live_src + (f"; committed pass (profiles/): {committed}" if committed else "")
            else:
                traffic_src = f"{traffic_src}; live pass failed: {live_src}"
        line = {
            "metric": "Mpixels/s remapped, 8K equirect->equidistant" if cfg["pin"] == "c2" else f"Mpixels/s remapped ({args.config})",
            "value": round(value, 1),
            "unit": "Mpx/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": round(dt_max / K * 1e3, 5),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",  # (per-tile float32 coordinate models, every index certified against the float64 chain at plan creation; u8 samples)
            "data": "synthetic",
            "config": {
                "workload": cfg["text"],
                "name": args.config,
                "frames_per_launch": batch,
                "frames_resident_per_gpu": pool,
                "streams": n_streams,
                "sampling": "bilinear, 4 taps (opt-in mode, not the reference's sampler)" if bilinear else "nearest (truncating), the reference's",
                "parallelism": f"frames sharded over {world} GPU(s); RCCL broadcast of the parameter block only",
            },
            "plan_create_ms": round(plan_create_ms, 3),
            "plan_create_warm_ms": round(plan_create_warm_ms, 3),
            "plan_timing": {k: round(v, 3) for k, v in plan.timing().items()},
            "first_frame_ms": round(first_frame_ms, 3),
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "traffic_live": bool(world == 1 and not args.no_live_traffic and traffic is not None and str(traffic_src).startswith("live")),
                "copy_ceiling_gbs": round(ceiling, 1),
                "frac_of_copy_ceiling": round(achieved / ceiling, 4),
                "algorithmic_bytes_per_launch": alg_launch,
                "must_move_bytes_per_launch": must_frame * batch,
                "attainable_frac": round(alg_frame / must_frame, 4),
                "attainable_frac_at_copy_ceiling": round(alg_frame / must_frame * ceiling / HBM_PEAK_GBS, 4),
                "kernel_ms_mean": round(launch_ms, 5),
                "kernel_ms_median": round(float(np.median(durs)), 5) if durs else None,
                "kernel_ms_p10": round(float(np.percentile(durs, 10)), 5) if durs else None,
                "kernel_ms_p90": round(float(np.percentile(durs, 90)), 5) if durs else None,
                "kernel_ms_per_frame": round(launch_ms / batch, 5),
                "timing": f"hipEvent pairs around groups of {every} consecutive timed launches on the launch stream ({len(groups)} groups; duration = group time / group size; one pb_remap_u8 call = ONE kernel launch incl. its fix work)" if durs else "wall / steps",
                "window_budget": info["window_budget"],
                "plan": info,
            },
        }
        line["ranks_seen"] = dist.get_world_size() if dist.is_initialized() else 1
        line["collective_backend"] = (dist.get_backend() if dist.is_initialized() else None)
        if world == 1 and not bilinear and not args.no_configs and args.config == "c2":
            # the other BASELINE configs, same process, same box, seconds: one driver-run line shows every kernel
            try:
                line["graph_replay_ms_per_frame"] = round(graph_replay_ms(lib, nat, plan, srcs, dsts, sbytes, dbytes, pool, device), 5)
                line["graph_replay_note"] = "8 single-frame launches captured into one hipGraph, replayed 25 times (same kernels, same frames of the pool): launch boundaries inside a graph cost what eager launches cost"
            except Exception as exc:  # a measurement extra: never fail the line over it
                line["graph_replay_ms_per_frame"] = None
                line["graph_replay_note"] = repr(exc)
            try:
                line["wall_ms_per_frame_by_streams"] = {str(n): round(multi_stream_ms(remap_fn, h, sp0, dp0, sbytes, dbytes, pool, batch, device, n), 5) for n in (1, 2, 3, 4)}
                line["wall_ms_per_frame_by_streams_note"] = ("120 independent single-frame launches dealt round-robin to N HIP streams, wall clock between two device syncs: "
                                                             "with N >= 2 the next launch's ramp runs in the previous one's drain (the C ABI takes the stream per call; "
                                                             "`value` and `roofline` above are the single-stream figures, where a per-kernel duration is defined)")
            except Exception as exc:
                line["wall_ms_per_frame_by_streams"] = None
                line["wall_ms_per_frame_by_streams_note"] = repr(exc)
            del srcs, dsts
            torch.cuda.empty_cache()
            try:
                line["scattered_batch"] = scattered_batch(lib, nat, plan, cfg, s, d, device, sts[0])
            except Exception as exc:
                line["scattered_batch"] = {"error": repr(exc)}
            single_ms, faithful_ms, bil_prepare_ms = headline_extras(lib, nat, d, rots, s, cfg, device, sts[0], budget)
            line["single_image_ms"] = single_ms
            line["plan_bilinear_prepare_ms"] = bil_prepare_ms
            line["single_image_note"] = "warm plan creation (thresholds, tile models, certification, launch table) + the first frame of a NEW geometry, device-resident input; a deferred plan instead runs faithful_kernel_ms with no preparation"
            line["faithful_kernel_ms"] = faithful_ms
            # the un-amortised figures travel WITH the headline: `frac` above prices a launch of a prepared plan;
            # one image of a new geometry pays either the preparation + one launch, or the float64 chain per pixel
            line["roofline"]["frac_unamortised"] = {
                "single_image": round(alg_frame / (single_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "faithful_kernel": round(alg_frame / (faithful_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "single_image_ms": single_ms, "faithful_kernel_ms": faithful_ms,
                "break_even_frames": round(max(0.0, single_ms - launch_ms) / max(1e-9, faithful_ms - launch_ms), 1),
                "note": "algorithmic bytes of ONE frame / (warm plan preparation + first launch) resp. / the float64 kernel's time, over 8 TB/s: what a caller who remaps a geometry once gets",
            }
            block = {}
            for name in ("c1", "c3", "c5", "c4shard", "c5shard"):
                block[name] = measure_config(lib, nat, name, device, sts[0])
            for name in ("c1", "c2", "c3", "c5"):
                block[name + "_bilinear"] = measure_config(lib, nat, name, device, sts[0], bilinear=True)
            line["configs"] = block
            line["flavours"] = flavours_block("c3")
            try:
                line["host_path"] = host_path(cfg, d, rots, s)
                line["host_path"]["cpus_after_numa_pin"] = numa_cpus  # (0: affinity left as found - one node, or already inside the GPU's)
            except Exception as exc:  # a measurement extra: never fail the line over it
                line["host_path"] = {"error": repr(exc)}
    # ---- north_star's multi-GPU workloads, at ANY rank count: c4 = 64 frames per GPU, c5 = 32 per GPU, sharded
    # with shard_range, 8 frames per launch; aggregate Mpx/s over all ranks; every rank's first output frame is re-made by rank 0
    # from the same frame index and must be byte-identical (SURVEY 8 e's acceptance check)
    sharded = None
    if not bilinear and not args.no_configs and args.config == "c2":
        import hashlib

        try:
            del srcs, dsts
        except NameError:
            pass
        torch.cuda.empty_cache()
        sharded = {}
        shard_cap = int(os.environ.get("PB_SHARD_FRAMES", "0"))  # rehearsals on a shared card: fewer frames per rank than BASELINE's 64 / 32 (0: BASELINE's)
        for wname, cname, per_rank in (("c4", "c2", 64), ("c5", "c5", 32)):
            if shard_cap > 0:
                per_rank = min(per_rank, shard_cap)
            passes = 3
            dt_w, n_mine, digest, first_k, wplan, (wd, ws, wcfg) = sharded_workload(lib, nat, parallel, cname, per_rank, rank, world, device, coll_device, sts[0], dist, passes=passes)
            tw = torch.tensor([dt_w], dtype=torch.float64, device=coll_device)
            mine_t = torch.tensor(list(digest) + [first_k & 0xFF, (first_k >> 8) & 0xFF, (first_k >> 16) & 0xFF, 0], dtype=torch.uint8, device=coll_device)
            gathered = [mine_t]
            if dist.is_initialized():
                dist.all_reduce(tw, op=dist.ReduceOp.MAX)
                gathered = [torch.empty_like(mine_t) for _ in range(world)]
                dist.all_gather(gathered, mine_t)
            if rank == 0:
                same = 0
                for r, g in enumerate(gathered):
                    gb = bytes(g.cpu().numpy().tolist())
                    k = gb[32] | (gb[33] << 8) | (gb[34] << 16)
                    assert k == parallel.shard_range(per_rank * world, world, r).start
                    frame = nat.synth_frame(ws.height, ws.width, frame=k, seed=0, circle_mask=wcfg["mask"])
                    mine_out = wplan.remap(frame)
                    if hashlib.sha256(mine_out.cpu().numpy().tobytes()).digest() != gb[:32]:
                        raise SystemExit(f"bench.py: {wname}: frame {k} remapped by rank {r} differs from rank 0's remap of the same frame")
                    same += 1
                    del frame, mine_out
                mpx = wd.height * wd.width / 1e6
                frames_total = world * per_rank * passes
                sharded[wname] = {
                    "workload": f"BASELINE config {wname[1]}: {per_rank * world} distinct frames of {CONFIGS[cname]['text'].split(':')[0]}'s geometry, {per_rank} per GPU resident in HBM, 8 per launch, sharded by parallel.shard_range",
                    "frames_total": per_rank * world, "frames_per_gpu": per_rank, "frames_per_launch": 8, "passes_timed": passes,
                    "mpx_per_s": round(frames_total * mpx / float(tw.item()), 1),
                    "ms_per_frame_per_gpu": round(float(tw.item()) * 1e3 / (per_rank * passes), 5),
                    "first_frames_identical_to_rank0": same,
                    "timing": "barrier + device sync, `passes` passes over each rank's share, device sync; MAX over ranks",
                }
            del wplan
            torch.cuda.empty_cache()
    if rank == 0:
        if sharded is not None:
            sharded["ranks_seen"] = dist.get_world_size() if dist.is_initialized() else 1
            sharded["collective_backend"] = dist.get_backend() if dist.is_initialized() else None
            sharded["collective"] = "one broadcast of the 90-double parameter block per workload (RCCL when the backend is nccl); no pixel crosses a link"
            line["sharded"] = sharded
        if world == 1 and not args.no_cpu_baseline and not bilinear:  # (the CPU leg times the reference's nearest sampler)
            if numa_cpus and cpus_as_found:
                os.sched_setaffinity(0, cpus_as_found)  # (the CPU legs run where the host would have put them: the pin is the GPU path's)
            line["cpu_baseline"] = cpu_baseline(cfg, mpx_per_frame)
            extra = cpu_baseline_all_cores(args.config, mpx_per_frame)
            if extra:
                line["cpu_baseline"]["all_cores"] = extra
        detail = args.detail or (os.path.join(ROOT, "gpurun_out", "bench_detail.json") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else None)
        if detail:
            try:
                with open(detail, "w") as fh:
                    json.dump(line, fh, indent=1)
            except OSError as exc:
                print(f"bench.py: could not write {detail}: {exc}", file=sys.stderr)
        if args.explain:
            print(json.dumps(line), file=sys.stderr, flush=True)
        text = json.dumps(compact_line(line), separators=(",", ":"))
        if len(text) >= LINE_LIMIT:
            print(f"bench.py: the line is {len(text)} bytes (limit {LINE_LIMIT})", file=sys.stderr)
        print(text, flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
