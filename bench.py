#!/usr/bin/env python
"""bench.py — throughput of the zignal image hot path on MI355X.

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
    Image.gaussianBlur(sigma=0.6)  ==  5x5 separable Gaussian, .mirror border,
    4096 x 4096 RGBA f32 (interleaved float4; per-channel results equal the reference's f32 plane path),
    frames resident in HBM before the timed region, rotating through a ring of distinct buffers whose
    footprint (>= 2 GiB) is far above the 256 MiB Infinity Cache so the rate is an HBM rate.

A step = one frame through the hot path (one kernel launch). N GPUs: one process per GPU, frames sharded
across ranks with no data-path collective (independent frames) -> weak scaling; value is the
whole-job Mpixels/s = N * frame pixels * steps / max-over-ranks time.

`python bench.py --gpus N` with N > 1 and no launcher around it starts its own ranks (one process per GPU through
torch.distributed.run on 127.0.0.1); under a launcher (RANK / WORLD_SIZE in the environment) it is one of them. The JSON says how
many ranks the RCCL process group really spanned (`ranks.rccl_ranks`: an all-reduce of 1) and every rank's own ms_per_step.

Extra legs (rank 0, N = 1 only):
    roofline      algorithmic bytes (32 B/px: 16 read + 16 written) / mean kernel time from HIP events
                  recorded around each launch on the launch stream, against the 8.0 TB/s HBM3E peak.
    cpu_baseline  the CPU oracle ("port" of the reference's convolveSeparablePlane, 1 thread) on a bounded
                  sample of the same workload.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

XGMI_LINK_GBS = 153.0  # per xGMI link and direction (MI355X: 7 links per GPU; the task statement's and SURVEY 8e's figure)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
PREWARM_S = 0.5  # untimed clock-ramp phase before the --warmup steps (see main)
ROWS = COLS = 4096
SIGMA = 0.6
METRIC = "Mpixels/s (and % HBM roofline), 5x5 blur + bilinear resize, 4K RGBA"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # the GPU's power management needs a few ms of sustained load to settle (kernel time drifts 90 -> 117 -> 87 us
    # over the first ~25 ms, profiles/r01_blur_rgba_f32_kernel_series.txt), hence the long default warm-up
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=400)
    ap.add_argument("--ring", type=int, default=4, help="distinct (src, dst) frame pairs to rotate through")
    ap.add_argument("--eager", action="store_true", help="launch every step from Python instead of replaying a HIP graph")
    ap.add_argument("--scatter-gather", action="store_true",
                    help="also time BASELINE configs[4] end to end (RCCL scatter -> blur+resize -> gather), 128 frames per GPU; "
                         "with one GPU the shard loops back through a one-rank RCCL communicator. With --gpus N > 1 this leg runs by default")
    ap.add_argument("--no-scatter-gather", action="store_true", help="N > 1: skip the configs[4] scatter / compute / gather leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="take roofline.traffic from profiles/traffic.json instead of two rocprofv3 counter passes")
    ap.add_argument("--no-extras", action="store_true")
    return ap.parse_args()


def _pin_to_one_core():
    """the CPU leg runs pinned (taskset -c): this process's affinity shrinks to one allowed core for the duration. Returns
    (previous affinity, core) or (None, None) where the platform has no affinity call."""
    try:
        before = os.sched_getaffinity(0)
        core = max(before)  # away from core 0, where the launcher and the HIP runtime's helper threads tend to sit
        os.sched_setaffinity(0, {core})
        return before, core
    except (AttributeError, OSError):
        return None, None


def parity_metrics(got, want):
    """PSNR and mean pixel error with the reference's formulas (src/image/metrics.zig:10-54, 114-171): every component as f64, the
    component maximum 255 for u8 and 1.0 for floats, PSNR = inf when the mean squared error is 0."""
    import numpy as np
    a, b = np.asarray(got, np.float64), np.asarray(want, np.float64)
    mx = 255.0 if np.asarray(got).dtype == np.uint8 else 1.0
    mse = float(np.mean((a - b) ** 2))
    psnr = float("inf") if mse == 0.0 else 20.0 * np.log10(mx) - 10.0 * np.log10(mse)
    return psnr, float(np.mean(np.abs(a - b)) / mx)


def cpu_baseline(budget_s: float = 12.0, gpu_blur=None):
    """Oracle ('port') timed on this box's host cores, one thread pinned to one core (taskset, best of 5 after a warm-up), on a
    bounded sample: whole 4096x4096 f32 planes (the reference has no Rgba(f32) convolution, so an RGBA f32 frame is four Image(f32) planes)."""
    import numpy as np
    from oracle import pyoracle as oracle  # checker / baseline only — never on the product path

    try:
        oracle.lib(native=True)
        native = True
    except Exception:
        native = False
    k = oracle.gaussian_kernel(SIGMA)
    plane = oracle.synth_f32(2, (ROWS, COLS))
    out = np.empty_like(plane)
    before, core = _pin_to_one_core()
    try:
        oracle.conv_separable(plane, k, k, oracle.MIRROR, out=out, native=native)  # the warm-up run (page faults, clocks)
        times = []
        t_start = time.perf_counter()
        while len(times) < 5 and (not times or (time.perf_counter() - t_start) < budget_s):
            t0 = time.perf_counter()
            oracle.conv_separable(plane, k, k, oracle.MIRROR, out=out, native=native)
            times.append(time.perf_counter() - t0)
    finally:
        if before is not None:
            os.sched_setaffinity(0, before)
    best = min(times)
    mpix = ROWS * COLS / (4 * best) / 1e6  # an RGBA frame = 4 planes
    parity = None
    if gpu_blur is not None:  # the same plane through the product: the reference's own image metrics between the two results
        psnr, mpe = parity_metrics(gpu_blur(plane), out)
        parity = {"psnr_db": "inf" if psnr == float("inf") else round(psnr, 2), "mean_pixel_error": mpe,
                  "formulas": "src/image/metrics.zig:10-54, 114-171", "sample": "the timed 4096x4096 f32 plane, GPU result vs CPU port"}
    return {"value": round(mpix, 2), "unit": "Mpixels/s", "cores": 1, "kind": "port", "parity_gpu_vs_cpu_port": parity,
            "sample": f"best of {len(times)} x gaussianBlur(0.6) on one 4096x4096 f32 plane after one warm-up run, x4 planes per RGBA frame; "
                      f"restated zignal CPU path (C, oracle/conv.c {'-march=native' if native else '-march=x86-64-v3'} -O3 -ffp-contract=off), not Zig-compiled; "
                      f"{'pinned to core %d (sched_setaffinity)' % core if core is not None else 'not pinned (no affinity call here)'}; "
                      f"host has {os.cpu_count()} logical cores"}


def launch_ranks(args) -> int:
    """--gpus N without a launcher: become the launcher. One process per GPU, rendezvous on 127.0.0.1 (the container's host name
    may not resolve), same arguments; rank 0's JSON line is the only thing the children print on stdout."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL's peer-to-peer set-up needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse_args()
    if args.gpus > 1 and not args.no_scatter_gather:
        args.scatter_gather = True  # the N-GPU run is the one that can put bytes on xGMI: BASELINE configs[4] is part of it unless switched off
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    # stdout carries exactly one line, the JSON: everything libraries print on the way (RCCL's version banner, for one) goes to stderr
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    import numpy as np
    import torch

    import zignal_amd as zg

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # ZG_BENCH_SHARED_GPU=1 (test hook, never set by the driver): every rank uses cuda:0 and the few control-plane
    # collectives run over gloo, so the N > 1 code path can be exercised on a one-GPU box. Rates measured that way mean nothing.
    shared_gpu = os.environ.get("ZG_BENCH_SHARED_GPU") == "1"
    if shared_gpu:
        local_rank = 0
    if world > 1 or args.scatter_gather:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:  # a one-GPU box can still execute the RCCL path: a one-rank communicator, shards looped back through it
            import socket
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                os.environ.setdefault("MASTER_PORT", str(sock.getsockname()[1]))
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        torch.cuda.set_device(local_rank)
        if shared_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if world != args.gpus:
        sys.exit(f"bench.py: --gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE)")
    if not shared_gpu and torch.cuda.device_count() <= local_rank:
        sys.exit(f"bench.py: rank {rank} wants cuda:{local_rank} but {torch.cuda.device_count()} GPUs are visible")
    torch.cuda.set_device(local_rank)
    lib = zg.lib()
    rc = lib.zg_init(local_rank)
    assert rc == 0, lib.zg_last_error()

    # synthetic frames, seeded per rank; uniform [0,1) f32 like SURVEY §8d
    gen = torch.Generator(device="cuda")
    gen.manual_seed(2 + rank)
    ring = max(2, args.ring)
    srcs = [torch.rand((ROWS, COLS, 4), dtype=torch.float32, device="cuda", generator=gen) for _ in range(ring)]
    dsts = [torch.empty_like(s) for s in srcs]
    imgs = [(zg.Image(s), zg.Image(d)) for s, d in zip(srcs, dsts)]
    ring_bytes = sum(s.numel() * 4 for s in srcs) * 2

    def step(i):
        s, d = imgs[i % ring]
        s.gaussian_blur(SIGMA, out=d)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # The launch-bound inner loop is captured once into a HIP graph (`ring` launches, one per ring slot) and
    # replayed; each replayed launch is still one frame through the hot path, and exactly `steps` of them are timed.
    graph = None
    side = torch.cuda.Stream()
    chunk = 0
    for cand in (25 * ring, 10 * ring, 5 * ring, ring):  # launches per graph: long enough to hide the ~11 us replay gap
        if args.steps % cand == 0:
            chunk = cand
            break
    if not args.eager and chunk:
        try:
            with torch.cuda.stream(side):
                for i in range(ring):
                    step(i)
                side.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph, stream=side):
                    for i in range(chunk):
                        step(i)
        except Exception as e:  # capture is an optimisation; eager launches are always valid
            print(f"[bench] graph capture failed ({e}); falling back to eager launches", file=sys.stderr)
            graph = None

    def run(n_steps):
        if graph is not None:
            for _ in range(n_steps // chunk):
                graph.replay()
        else:
            for i in range(n_steps):
                step(i)

    # Clock ramp: a cold MI355X takes a few hundred ms of sustained work to reach its running clocks, which a small
    # --warmup does not provide; this untimed phase is the same launches as the timed region and is reported in config.
    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < PREWARM_S:
        run(chunk if graph is not None else ring)
        torch.cuda.synchronize()
    run(max(chunk, args.warmup - args.warmup % chunk) if graph is not None else args.warmup)
    barrier()
    ev_begin, ev_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev_begin.record()  # HIP events on the stream the launches (or graph replays) go to, around exactly the timed region
    run(args.steps)
    ev_end.record()
    barrier()
    elapsed = time.perf_counter() - t0
    region_ms_per_launch = ev_begin.elapsed_time(ev_end) / args.steps
    from zignal_amd import sharding
    ctl = torch.device("cpu") if shared_gpu else torch.device("cuda", local_rank)  # where the control-plane collectives run
    own_ms = elapsed / args.steps * 1e3
    elapsed = sharding.max_over_ranks(elapsed, ctl)  # the slowest rank is the clock
    ranks = {"backend": "none (one process)", "rccl_ranks": None, "ms_per_step_per_rank": [round(own_ms, 5)], "devices": [torch.cuda.get_device_name(local_rank)]}
    if world > 1 or args.scatter_gather:
        seen = sharding.count_ranks(ctl)  # an all-reduce(SUM) of 1 over the process group: what the group spans, not what was asked for
        names = [None] * world
        dist.all_gather_object(names, f"cuda:{local_rank} {torch.cuda.get_device_name(local_rank)}")
        ranks = {"backend": "gloo (ZG_BENCH_SHARED_GPU test hook: every rank on cuda:0)" if shared_gpu else "nccl (RCCL)",
                 "rccl_ranks": None if shared_gpu else seen, "ranks_seen": seen,
                 "ms_per_step_per_rank": [round(v, 5) for v in sharding.per_rank(own_ms, ctl)], "devices": names}

    pixels = ROWS * COLS
    value = world * pixels * args.steps / elapsed / 1e6
    result = {
        "metric": METRIC, "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic", "ranks": ranks,
        "config": {"workload": "gaussianBlur(sigma=0.6): 5x5 separable Gaussian, mirror border, 4096x4096 RGBA f32 "
                               "(BASELINE.json configs[1]); one frame per step per GPU, frames resident in HBM",
                   "frame": [ROWS, COLS, 4], "ring_bytes": ring_bytes, "frames_per_step_per_gpu": 1, "untimed_clock_ramp_s": PREWARM_S,
                   "launch": f"hipGraph replay, {chunk} launches per graph" if graph is not None else "eager",
                   "parallelism": f"frame-sharded x{world}, no data-path collective"},
    }

    if rank == 0:
        # Kernel time per launch = HIP events around the timed region (above) / launches in it: device time on the launch stream,
        # inter-launch gaps included, so it can only overstate the kernel. Cross-check: event pairs around 200 single eager
        # launches (each pair also brackets its launch overhead), and the rocprofv3 mean under profiles/.
        # With N > 1 this is rank 0's own GPU (the ranks run the same launches on frames of their own), so the N = 1 line and
        # SCALE's first point agree by construction; the counter passes (a second process on the GPU) are an N = 1 matter.
        n = min(args.steps, 200)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for i, (a, b) in enumerate(evs):
            a.record()
            step(i)
            b.record()
        torch.cuda.synchronize()
        kernel_ms = sorted(a.elapsed_time(b) for a, b in evs)
        eager_mean_ms = sum(kernel_ms) / len(kernel_ms)
        mean_ms = region_ms_per_launch
        alg_bytes = 32 * pixels  # 16 B read + 16 B written per pixel
        achieved = alg_bytes / (mean_ms * 1e-3) / 1e9
        live = world == 1 and not args.no_live_traffic
        traffic, traffic_source = (live_traffic("blur_f32", "k_sep_fused") if live else None), "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, this run"
        if traffic is None:
            traffic_source = "profiles/traffic.json (an earlier rocprofv3 measurement of the same kernel)"
            try:
                traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("k_sep_fused_rgba_f32_bytes_per_launch")
            except Exception:
                traffic, traffic_source = None, None
        result["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source,
                              "kernel": "k_sep_fused<RGBA_F32,5>", "kernel_ms_mean": round(mean_ms, 5),
                              "kernel_ms_source": "HIP events around the timed region / launches in it" + (" (rank 0's GPU)" if world > 1 else ""),
                              "kernel_ms_eager_event_pairs_mean": round(eager_mean_ms, 5),
                              "kernel_ms_eager_event_pairs_median": round(kernel_ms[len(kernel_ms) // 2], 5),
                              "algorithmic_bytes_per_launch": alg_bytes}
    if rank == 0 and world == 1:
        # The metric names two ops: the bilinear resize of BASELINE configs[2] stands beside the blur, same arithmetic.
        try:
            result["resize"] = resize_headline(zg, torch, not args.no_live_traffic)
        except Exception as e:  # never take the headline down
            result["resize"] = {"error": f"{type(e).__name__}: {e}"}
        if not args.no_extras:
            if args.no_live_traffic:
                os.environ["ZG_BENCH_NO_LIVE_TRAFFIC"] = "1"
            result["extras"] = extras(zg, torch, np)
        if not args.no_cpu_baseline:
            def gpu_blur(plane):
                out = zg.Image(torch.from_numpy(plane).cuda()).gaussian_blur(SIGMA)
                torch.cuda.synchronize()
                return out.to_numpy()
            result["cpu_baseline"] = cpu_baseline(gpu_blur=gpu_blur)
            try:
                result["cpu_baseline_config5_all_cores"] = cpu_config5_all_cores()
            except Exception as e:
                result["cpu_baseline_config5_all_cores"] = {"error": f"{type(e).__name__}: {e}"}
            if not args.no_extras:
                cpu_extras(result["extras"])

    if args.scatter_gather:
        # Not the headline: BASELINE configs[4] end to end — rank 0 holds the batch, shards fan out over RCCL/xGMI
        # (grouped send/recv), every rank runs blur+resize on its shard, results are gathered back.
        try:
            sg = scatter_gather_leg(zg, torch, sharding, rank, world, local_rank)
            if rank == 0:
                result["scatter_gather_config5"] = sg
        except Exception as e:
            if rank == 0:
                result["scatter_gather_config5"] = {"error": f"{type(e).__name__}: {e}"}
    if world > 1 or args.scatter_gather:
        dist.destroy_process_group()
    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()


def live_traffic(op: str, kernel, launches: int = 5):
    """HBM bytes per launch of `kernel` measured now, on this box: two rocprofv3 passes (--pmc FETCH_SIZE, then WRITE_SIZE, --kernel-trace
    only, as MI355X_MICROARCH.md prescribes) over tools/run_op.py <op>; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (the counters are in
    KiB, and gfx950's FETCH_SIZE counts 32-byte units as 64). kernel=None: every kernel of the library ("zg::") the op launches, summed and
    divided by the op's calls — the bytes ONE call of a multi-kernel op moves. None when rocprofv3 is not on the box or a pass fails."""
    import shutil
    import sqlite3
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    got = {}
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as d:
                env = dict(os.environ, TMPDIR="/tmp", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))  # run_op.py imports zignal_amd from the cwd otherwise
                p = subprocess.run([exe, "--kernel-trace", "--pmc", ctr, "-d", d, "-o", "r", "--", sys.executable, os.path.join(ROOT, "tools", "run_op.py"), op, str(launches)],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=240)
                dbs = [os.path.join(r, f) for r, _, fs in os.walk(d) for f in fs if f.endswith(".db")]
                if p.returncode != 0 or not dbs:
                    return None
                vals = [v for k, name, v in sqlite3.connect(dbs[0]).execute("select kernel_name, counter_name, value from counters_collection")
                        if (kernel in k if kernel else "zg::" in k) and name == ctr]
                if not vals:
                    return None
                got[ctr] = sum(vals) / (len(vals) if kernel else launches)
        return int((2 * got["FETCH_SIZE"] + got["WRITE_SIZE"]) * 1024)
    except Exception:
        return None


def resize_headline(zg, torch, live=True):
    """Image(Rgba(u8)).resize(.bilinear) 4096^2 -> 1024^2 (BASELINE configs[2], channel_ops.zig:144-190), sources rotating through
    1 GiB of distinct frames. Strict algorithmic bytes (20 B per output pixel) and, beside them, what DRAM has to move for this
    geometry: the taps are bytes 4..11 of every 16 in rows 1, 2 mod 4 — every 64-byte line of half the rows (counter-checked:
    profiles/r02_pmc_traffic.txt), which caps the strict fraction at 0.556 of the bandwidth reached."""
    ring = 16
    srcs = [torch.randint(0, 256, (ROWS, COLS, 4), dtype=torch.uint8, device="cuda") for _ in range(ring)]
    im = [(zg.Image(s), zg.Image(torch.empty((1024, 1024, 4), dtype=torch.uint8, device="cuda"))) for s in srcs]
    ms = _time_kernel(torch, lambda i: im[i % ring][0].resize(im[i % ring][1], zg.Interpolation.bilinear), n=64, warm=8)
    alg = 20 * 1024 * 1024
    traffic = live_traffic("resize", "k_resize_bilinear_rgba8") if live else None
    if traffic is None:
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get("k_resize_bilinear_rgba8_4096_to_1024_bytes_per_launch")
        except Exception:
            pass
    dram = ROWS * COLS * 4 // 2 + 4 * 1024 * 1024
    achieved = alg / (ms * 1e-3) / 1e9
    # the same resize as a step of zg_batch_pipeline over 16 frames per launch (one frame is 4 096 one-gather workgroups: launch ramp and tail)
    nb = 16
    del im, srcs
    batches = [torch.randint(0, 256, (nb, ROWS, COLS, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]  # 2 GiB of sources
    bouts = [torch.empty((nb, 1024, 1024, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    pipe = zg.Pipeline([zg.Step.resize(1024, 1024)])
    bms = _time_kernel(torch, lambda i: pipe.run(batches[i % 2], out=bouts[i % 2]), n=8, warm=2) / nb
    batched = {"frames_per_launch": nb, "ms_per_frame": round(bms, 5), "Mpixels/s_source": round(ROWS * COLS / bms / 1e3, 1),
               "frac_strict": round(alg / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "frac_on_dram_granular_bytes": round(dram / (bms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
               "kernel": "k_resize_bilinear_rgba8<1>, frame index in the grid (zg_batch_pipeline)"}
    return {"workload": "resize(.bilinear) 4096x4096 -> 1024x1024 Rgba(u8), BASELINE.json configs[2]; one frame per launch, graph-replayed", "batched": batched,
            "ms_per_step": round(ms, 5), "Mpixels/s_source": round(ROWS * COLS / ms / 1e3, 1), "Mpixels/s_output": round(1024 * 1024 / ms / 1e3, 1),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "kernel": "k_resize_bilinear_rgba8<1>", "algorithmic_bytes_per_launch": alg,
                         "dram_granular_bytes_per_launch": dram, "frac_on_dram_granular_bytes": round(dram / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                         "strict_frac_ceiling_of_this_geometry": round(alg / dram, 3)}}


def cpu_config5_all_cores(budget_s: float = 20.0):
    """config 5 as an N-way frame-parallel CPU run (the reference has no threading of its own: N
    independent callers, one 1080p frame each at a time). Oracle port, N = the host's logical cores (capped at 64)."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    from oracle import pyoracle as oracle  # baseline leg only

    try:
        oracle.lib(native=True)
        native = True
    except Exception:
        native = False
    n = max(1, min(os.cpu_count() or 1, 64))
    bil = oracle.method(oracle.BILINEAR)
    frames = [oracle.synth_u8(5 + i, (1080, 1920, 4)) for i in range(min(n, 8))]

    def one(i):  # ctypes releases the GIL for the duration of the C calls
        return oracle.resize(oracle.gaussian_blur(frames[i % len(frames)], SIGMA, native=native), (540, 960), bil)

    t0 = time.perf_counter(); one(0); single = time.perf_counter() - t0
    per_thread = max(1, min(8, int(budget_s / 2 / max(single, 1e-3))))
    with ThreadPoolExecutor(n) as ex:
        list(ex.map(one, range(n)))  # warm-up: page faults, thread start
        t0 = time.perf_counter()
        list(ex.map(one, range(n * per_thread)))
        wall = time.perf_counter() - t0
    return {"value": round(n * per_thread * 1080 * 1920 / wall / 1e6, 1), "unit": "Mpixels/s", "cores": n, "kind": "port",
            "one_caller_Mpixels/s": round(1080 * 1920 / single / 1e6, 1),
            "sample": f"{n * per_thread} x [gaussianBlur(0.6), resize(.bilinear, 540x960)] on 1080p Rgba(u8) frames, {n} concurrent callers"}


def scatter_gather_leg(zg, torch, sharding, rank, world, local_rank, frames_per_gpu=128, chunks=4):
    """BASELINE configs[4] three ways, 128 frames of 1080p Rgba(u8) per GPU through [gaussianBlur(0.6), resize 0.5 bilinear]:
    (i)   kernel only: every rank's shard already resident, one batched launch per rank;
    (ii)  end to end over xGMI: rank 0 holds the whole batch, shards go out and results come back over RCCL point-to-point, cut into
          `chunks` pieces so that transfer, compute and the return trip overlap (sharding.scatter_compute_gather);
    (iii) host-staged: every rank feeds its own GPU from pinned host memory over its own PCIe link and takes the results back, chunked.
    Rates are whole-job: all ranks' source pixels / the slowest rank's time."""
    import ctypes as C
    import torch.distributed as dist
    dev = torch.device("cuda", local_rank)
    shared = os.environ.get("ZG_BENCH_SHARED_GPU") == "1"
    ctl = torch.device("cpu") if shared else dev
    n, rows, cols = frames_per_gpu * world, 1080, 1920
    lib = zg.lib()
    m = zg.Interpolation.bilinear._c()
    px_all = n * rows * cols

    def blur_resize(src, dst):  # on the current stream
        rc = lib.zg_batch_blur_resize(C.c_void_p(src.data_ptr()), int(src.shape[0]), rows, cols, 3, C.c_float(SIGMA), C.c_void_p(dst.data_ptr()),
                                      540, 960, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, lib.zg_last_error()

    def clock(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dist.barrier()
        return sharding.max_over_ranks((time.perf_counter() - t0) / reps, ctl)

    out = {"frames": n, "frames_per_gpu": frames_per_gpu, "chunks": chunks, "backend": "gloo (test hook)" if shared else "nccl (RCCL)"}
    # (i) resident shards
    shard = torch.randint(0, 256, (frames_per_gpu, rows, cols, 4), dtype=torch.uint8, device=dev)
    res = torch.empty((frames_per_gpu, 540, 960, 4), dtype=torch.uint8, device=dev)
    sec = clock(lambda: blur_resize(shard, res), reps=5)
    out["kernel_only"] = {"seconds": round(sec, 6), "Mpixels/s": round(px_all / sec / 1e6, 1)}

    # (iii) host-staged, chunked: copy-in stream -> compute stream -> copy-out stream
    host_in = torch.empty((frames_per_gpu, rows, cols, 4), dtype=torch.uint8).pin_memory()
    host_in.copy_(shard.cpu())
    host_out = torch.empty((frames_per_gpu, 540, 960, 4), dtype=torch.uint8).pin_memory()
    s_in, s_k, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    pieces = sharding.chunk_ranges(frames_per_gpu, 2 * chunks)

    def host_staged():
        evs = []
        for c0, c1 in pieces:
            with torch.cuda.stream(s_in):
                shard[c0:c1].copy_(host_in[c0:c1], non_blocking=True)
                e_in = torch.cuda.Event(); e_in.record()
            with torch.cuda.stream(s_k):
                s_k.wait_event(e_in)
                blur_resize(shard[c0:c1], res[c0:c1])
                e_k = torch.cuda.Event(); e_k.record()
            with torch.cuda.stream(s_out):
                s_out.wait_event(e_k)
                host_out[c0:c1].copy_(res[c0:c1], non_blocking=True)
            evs.append((e_in, e_k))
        s_out.synchronize()
    sec = clock(host_staged)
    out["host_staged_pcie"] = {"seconds": round(sec, 6), "Mpixels/s": round(px_all / sec / 1e6, 1),
                               "GB/s_per_gpu_in": round(frames_per_gpu * rows * cols * 4 / sec / 1e9, 1), "pieces": len(pieces)}
    del host_in, host_out

    # (ii) over the fabric: RCCL point-to-point, pipelined
    if shared:
        out["end_to_end_xgmi"] = {"skipped": "every rank shares cuda:0 under the test hook; RCCL needs one GPU per rank"}
    else:
        loop = world == 1  # one rank: its shard goes through ncclSend / ncclRecv to itself instead of staying in place
        second = dist.new_group(backend="nccl")  # results return on a communicator (and stream) of their own
        batch = torch.randint(0, 256, (n, rows, cols, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
        gathered = torch.empty((n, 540, 960, 4), dtype=torch.uint8, device=dev) if rank == 0 else None
        for label, k in (("end_to_end_xgmi", chunks), ("end_to_end_xgmi_unchunked", 1)):
            sec = clock(lambda: sharding.scatter_compute_gather(batch, n, (rows, cols, 4), (540, 960, 4), torch.uint8, dev, blur_resize, chunks=k,
                                                                loopback=loop, gather_group=second, out=gathered))
            sent = (n - (0 if loop else frames_per_gpu)) * rows * cols * 4  # bytes that leave rank 0 (a quarter as many come back)
            links = max(1, world - 1)
            # SURVEY 8e's bound: every peer's shard on its own xGMI link in parallel (XGMI_LINK_GBS per link and direction), the results (a quarter of the
            # bytes) coming back on the links' other direction at the same time: the scatter of one shard is the floor of the whole exchange
            per_link_bytes = sent / links
            bound_s = per_link_bytes / (XGMI_LINK_GBS * 1e9)
            out[label] = {"seconds": round(sec, 6), "Mpixels/s": round(px_all / sec / 1e6, 1), "GB/s_scattered": round(sent / sec / 1e9, 1),
                          "GB/s_scattered_per_link": round(per_link_bytes / sec / 1e9, 1), "links": links,
                          "xgmi_bound": None if loop else {"per_link_GB/s": XGMI_LINK_GBS, "bytes_per_link": int(per_link_bytes), "seconds": round(bound_s, 6),
                                                           "frac_of_bound": round(bound_s / sec, 4),
                                                           "meaning": "time to move one peer's shard over its own link at the link's rate / the measured end-to-end time "
                                                                      "(1.0 = transfer-bound with compute and the return trip fully hidden)"}}
        out["note"] = ("one rank: the shard loops back through the communicator (ncclSend / ncclRecv to itself), so this times RCCL's "
                       "device-local copy path, not xGMI" if world == 1 else
                       "rank 0 sends 8.3 MB per frame to the frame's owner and receives 2.1 MB back, each peer over its own xGMI link")
    return out


def cpu_extras(extras_out):
    """The oracle ('port', one host thread) on a bounded sample of each extra configuration, so every GPU figure has the
    reference's CPU path beside it. Samples are strips of the same workload; rates are per source pixel like the GPU's."""
    import numpy as np
    from oracle import pyoracle as oracle  # baseline leg only

    def timed(fn, px):
        fn()
        t0 = time.perf_counter()
        fn()
        return round(px / (time.perf_counter() - t0) / 1e6, 2)

    def put(name, value, sample):
        if name in extras_out and isinstance(extras_out[name], dict):
            extras_out[name]["cpu_port_Mpixels/s"] = value
            extras_out[name]["cpu_sample"] = sample

    try:
        native = True
        oracle.lib(native=True)
    except Exception:
        native = False
    bil = oracle.method(oracle.BILINEAR)
    bic = oracle.method(oracle.BICUBIC)
    small = oracle.synth_u8(1, (256, 256))
    put("config1_box_blur_256_u8", timed(lambda: oracle.box_blur(small, 1), 256 * 256), "the whole 256 x 256 frame")
    u8 = oracle.synth_u8(2, (512, COLS, 4))
    put("config2b_gaussian_blur_rgba_u8_4096", timed(lambda: oracle.gaussian_blur(u8, SIGMA, native=native), 512 * COLS), "512 x 4096 strip")
    src = oracle.synth_u8(3, (1024, COLS, 4))
    put("config3_resize_bilinear_rgba_u8_4096_to_1024", timed(lambda: oracle.resize(src, (256, 1024), bil), 1024 * COLS), "1024 x 4096 -> 256 x 1024")
    put("config3_convert_rgba_u8_to_oklab_f32_4096", timed(lambda: oracle.convert(u8, oracle.CS_RGBA, oracle.CS_OKLAB, np.float32, 3), 512 * COLS), "512 x 4096 strip")
    hmat = oracle.homography_from_4pts([(0, 0), (4095, 0), (0, 4095), (4095, 4095)], [(200, 120), (3900, 60), (90, 3980), (4000, 4050)])
    full_u8 = oracle.synth_u8(4, (ROWS, COLS, 4))
    put("config4_warp_projective_bicubic_rgba_u8_4096", timed(lambda: oracle.warp(full_u8, (128, COLS), oracle.PROJECTIVE, hmat, bic), 128 * COLS), "first 128 output rows")
    full_f32 = oracle.synth_f32(4, (ROWS, COLS, 4))
    put("config4_warp_projective_bicubic_rgba_f32_4096", timed(lambda: oracle.warp(full_f32, (128, COLS), oracle.PROJECTIVE, hmat, bic), 128 * COLS), "first 128 output rows")
    frame = oracle.synth_u8(5, (1080, 1920, 4))
    put("config5_batch_blur_resize_64x1080p_rgba_u8",
        timed(lambda: oracle.resize(oracle.gaussian_blur(frame, SIGMA, native=native), (540, 960), bil), 1080 * 1920), "one 1080p frame")


def _time_kernel(torch, fn, n=50, warm=5, capture=True):
    """Mean ms per call of fn(i): the n calls are captured into one HIP graph and replayed (so microsecond kernels are
    not timed through Python launch overhead); eager launches if capture is not possible."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    graph = None
    try:
        if not capture:
            raise RuntimeError("eager timing requested")
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn(0)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                for i in range(n):
                    fn(i)
        graph.replay()
        torch.cuda.synchronize()
    except Exception:
        graph = None
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 3 if graph is not None else 1
    a.record()
    for _ in range(reps):
        if graph is not None:
            graph.replay()
        else:
            for i in range(n):
                fn(i)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / (n * reps)
    if graph is not None:  # scratch that captured calls took stays with the graph until it is given back
        del graph
        import zignal_amd as _zg
        _zg.lib().zg_release_graph_scratch()
    return ms


def _time_kernel_two_streams(torch, fn, n=50, warm=5):
    """As _time_kernel, but the n calls of the captured graph alternate between TWO streams forked from the capture stream and joined back at the end
    (VERDICT r05 next-6): independent frames, still one frame per launch, so that one launch's ramp overlaps its predecessor's tail instead of waiting
    behind it on the same stream. Mean ms per call; None if the capture fails."""
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    try:
        side, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        with torch.cuda.stream(side):
            fn(0)
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                s1.wait_stream(side)
                s2.wait_stream(side)
                for i in range(n):
                    with torch.cuda.stream(s1 if i % 2 == 0 else s2):
                        fn(i)
                side.wait_stream(s1)
                side.wait_stream(s2)
        graph.replay()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3):
            graph.replay()
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / (n * 3)
        del graph
        import zignal_amd as _zg
        _zg.lib().zg_release_graph_scratch()
        return ms
    except Exception:
        return None


def extras(zg, torch, np):
    """Secondary numbers (not the headline): the other configurations of BASELINE.json, each with its
    algorithmic bytes so the same roofline arithmetic applies. Ring-rotated buffers, HIP events."""
    out = {}
    I = zg.Interpolation

    only = os.environ.get("ZG_BENCH_EXTRAS")  # a regular expression: run only the legs whose name matches (A/B runs of library variants)

    def leg(name, fn):
        if only and not re.search(only, name):
            return
        try:
            out[name] = fn()
        except Exception as e:  # an extra must never take the headline down
            out[name] = {"error": f"{type(e).__name__}: {e}"}

    def rate(ms, px, bytes_alg):
        return {"ms": round(ms, 5), "Mpixels/s": round(px / ms / 1e3, 1), "GB/s_algorithmic": round(bytes_alg / ms / 1e6, 1),
                "frac_of_8TB/s": round(bytes_alg / ms / 1e6 / HBM_PEAK_GBS, 4)}

    def two_streams(r, fn):
        """the same leg with the launches alternating between two forked streams (a separate key: the single-stream figure stays the leg's own)"""
        ms2 = _time_kernel_two_streams(torch, fn)
        if ms2 is not None:
            r["two_streams_ms"] = round(ms2, 5)
        return r

    def u8_frames(n, shape):
        return [torch.randint(0, 256, shape, dtype=torch.uint8, device="cuda") for _ in range(n)]

    def blur_u8():
        ring = 8
        im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].gaussian_blur(SIGMA, out=im[i % ring][1]))
        return two_streams(rate(ms, ROWS * COLS, 8 * ROWS * COLS), lambda i: im[i % ring][0].gaussian_blur(SIGMA, out=im[i % ring][1]))  # 4 B read + 4 B written per pixel

    def resize_u8():
        ring = 16  # 1 GiB of sources
        im = [(zg.Image(s), zg.Image(torch.empty((1024, 1024, 4), dtype=torch.uint8, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].resize(im[i % ring][1], I.bilinear))
        r = rate(ms, ROWS * COLS, 20 * 1024 * 1024)  # 4 taps x 4 B + 4 B per OUTPUT pixel (strict)
        r["GB/s_sector_basis"] = round((ROWS * COLS * 4 // 2 + 4 * 1024 * 1024) / ms / 1e6, 1)  # rows 1,2 mod 4 are touched whole
        return two_streams(r, lambda i: im[i % ring][0].resize(im[i % ring][1], I.bilinear))

    def resize_dense(sr, dr):
        # a dense case for the same kernel: every source byte is a tap (2:1) or every destination byte is new (1:2),
        # so the strict algorithmic bytes are what DRAM has to move
        ring = max(2, -(-(1 << 30) // (4 * sr * sr + 4 * dr * dr)))  # >= 1 GiB of distinct buffers: past the 256 MiB Infinity Cache
        im = [(zg.Image(torch.randint(0, 256, (sr, sr, 4), dtype=torch.uint8, device="cuda")), zg.Image(torch.empty((dr, dr, 4), dtype=torch.uint8, device="cuda")))
              for _ in range(ring)]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].resize(im[i % ring][1], I.bilinear), n=20, warm=3)
        return rate(ms, dr * dr, 4 * sr * sr + 4 * dr * dr)

    def oklab():
        ring = 8
        im = [(zg.Image(s), zg.Image(torch.empty((ROWS, COLS, 3), dtype=torch.float32, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].convert(zg.CS_OKLAB, np.float32, out=im[i % ring][1]))
        return rate(ms, ROWS * COLS, 16 * ROWS * COLS)  # 4 B read + 12 B written

    def resize_oklab_fused():
        # BASELINE configs[2] as ONE call: bilinear 4096^2 -> 1024^2 Rgba(u8) and Rgb -> Oklab(f32) fused (zg_resize_convert)
        ring = 16
        im = [(zg.Image(s), zg.Image(torch.empty((1024, 1024, 3), dtype=torch.float32, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].resize_convert(im[i % ring][1], zg.CS_OKLAB))
        r = rate(ms, ROWS * COLS, 28 * 1024 * 1024)  # 4 taps x 4 B read + 12 B written per OUTPUT pixel
        r["GB/s_sector_basis"] = round((ROWS * COLS * 4 // 2 + 12 * 1024 * 1024) / ms / 1e6, 1)
        return two_streams(r, lambda i: im[i % ring][0].resize_convert(im[i % ring][1], zg.CS_OKLAB))

    def warp(kind):
        tr = zg.ProjectiveTransform.from_points([(0, 0), (4095, 0), (0, 4095), (4095, 4095)], [(200, 120), (3900, 60), (90, 3980), (4000, 4050)])
        ring = 4
        if kind == "u8":
            srcs, bpp = u8_frames(ring, (ROWS, COLS, 4)), 8
        else:
            srcs, bpp = [torch.rand((ROWS, COLS, 4), dtype=torch.float32, device="cuda") for _ in range(ring)], 32
        im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in srcs]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].warp(tr, im[i % ring][1], I.bicubic), n=20, warm=3)
        return rate(ms, ROWS * COLS, bpp * ROWS * COLS)

    def batch(n=64):
        import ctypes as C
        rows, cols = 1080, 1920
        src = torch.randint(0, 256, (n, rows, cols, 4), dtype=torch.uint8, device="cuda")
        dst = torch.empty((n, 540, 960, 4), dtype=torch.uint8, device="cuda")
        m = I.bilinear._c()
        lib = zg.lib()

        def run(_):
            rc = lib.zg_batch_blur_resize(C.c_void_p(src.data_ptr()), n, rows, cols, 3, C.c_float(SIGMA), C.c_void_p(dst.data_ptr()),
                                          540, 960, C.byref(m), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, lib.zg_last_error()
        ms = _time_kernel(torch, run, n=10, warm=2)
        r = rate(ms, n * rows * cols, n * 10368000)  # fused bound: 8 294 400 read + 2 073 600 written per frame
        r["frames"] = n
        return r

    def blur_planes():
        ring = 8
        im = [(zg.Image(torch.rand((ROWS, COLS), dtype=torch.float32, device="cuda")), zg.Image(torch.empty((ROWS, COLS), dtype=torch.float32, device="cuda")))
              for _ in range(ring)]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].gaussian_blur(SIGMA, out=im[i % ring][1]))
        return two_streams(rate(ms, ROWS * COLS, 8 * ROWS * COLS), lambda i: im[i % ring][0].gaussian_blur(SIGMA, out=im[i % ring][1]))  # one f32 plane: 4 B read + 4 B written per pixel

    def blur_planes4():
        # BASELINE configs[1] in the one form a zignal caller can express for f32 data: four Image(f32) planes (convolveSeparable rejects
        # Rgba(f32) at comptime, convolution.zig:431-435), all four in ONE launch (zg_gaussian_blur_planes). 2 GiB of planes.
        ring = 4
        quads = [([zg.Image(torch.rand((ROWS, COLS), dtype=torch.float32, device="cuda")) for _ in range(4)],
                  [zg.Image(torch.empty((ROWS, COLS), dtype=torch.float32, device="cuda")) for _ in range(4)]) for _ in range(ring)]
        ms = _time_kernel(torch, lambda i: zg.gaussian_blur_planes(quads[i % ring][0], SIGMA, outs=quads[i % ring][1]), n=20, warm=4)
        r = rate(ms, ROWS * COLS, 32 * ROWS * COLS)  # per RGBA pixel: 16 B read + 16 B written
        r["planes_per_launch"] = 4
        return r

    def sobel():
        ring = 8
        im = [(zg.Image(s), zg.Image(torch.empty((ROWS, COLS), dtype=torch.uint8, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].sobel(out=im[i % ring][1]))
        return rate(ms, ROWS * COLS, 5 * ROWS * COLS)  # 4 B read + 1 B written per pixel (fused)

    def lab(forward):
        ring = 4
        if forward:
            im = [(zg.Image(s), zg.Image(torch.empty((ROWS, COLS, 3), dtype=torch.float32, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
            ms = _time_kernel(torch, lambda i: im[i % ring][0].convert(zg.CS_LAB, np.float32, out=im[i % ring][1]), n=20, warm=3)
        else:
            labs = [zg.Image(s).convert(zg.CS_LAB, np.float32) for s in u8_frames(ring, (ROWS, COLS, 4))]
            im = [(l, zg.Image(torch.empty((ROWS, COLS, 4), dtype=torch.uint8, device="cuda"))) for l in labs]
            ms = _time_kernel(torch, lambda i: im[i % ring][0].convert(zg.CS_RGBA, np.uint8, src_space=zg.CS_LAB, out=im[i % ring][1]), n=20, warm=3)
        return rate(ms, ROWS * COLS, 16 * ROWS * COLS)  # 4 B + 12 B per pixel either way

    def pyramid_blur():
        # the blur of ORB pyramid level 3 on a grey frame: sigma = 1.6 * sqrt(1.2^6 - 1) = 2.25 (15 taps), packed two-pass path
        ring = 4
        im = [(zg.Image(torch.randint(0, 256, (ROWS, COLS), dtype=torch.uint8, device="cuda")), zg.Image(torch.empty((ROWS, COLS), dtype=torch.uint8, device="cuda")))
              for _ in range(ring)]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].gaussian_blur(2.2528, out=im[i % ring][1]), n=20, warm=3)
        return rate(ms, ROWS * COLS, 2 * ROWS * COLS)

    def canny():
        # the whole detector (grey, Gaussian, Sobel, NMS, hysteresis by component labelling): nine launches, no host sync
        ring = 4
        im = [(zg.Image(s), zg.Image(torch.empty((ROWS, COLS), dtype=torch.uint8, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].canny(1.4, 50, 150, out=im[i % ring][1]), n=8, warm=2)
        return rate(ms, ROWS * COLS, 5 * ROWS * COLS)

    def shen():
        ring = 2
        im = [(zg.Image(s), zg.Image(torch.empty((ROWS, COLS), dtype=torch.uint8, device="cuda"))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].shen_castan(out=im[i % ring][1]), n=4, warm=1)
        return rate(ms, ROWS * COLS, 5 * ROWS * COLS)

    def photo_like():
        # smooth colour fields, a few hundred hard-edged discs, a little sensor noise: what the detectors usually see. The noise frames above
        # are their worst case (a quarter to 40 % of the pixels are candidates and the components percolate across the hysteresis tiles).
        g = torch.Generator(device="cuda").manual_seed(7)
        yy, xx = torch.meshgrid(torch.arange(ROWS, device="cuda"), torch.arange(COLS, device="cuda"), indexing="ij")
        pic = torch.stack([128 + 90 * torch.sin(xx / 310.0) * torch.cos(yy / 270.0), 128 + 80 * torch.cos(xx / 190.0 + yy / 400.0), 128 + 100 * torch.sin((xx + yy) / 520.0)], -1)
        for cx, cy, rad in torch.randint(0, min(ROWS, COLS), (300, 3), generator=g, device="cuda").tolist():
            m = ((xx - cx) ** 2 + (yy - cy) ** 2) < (20 + rad % 180) ** 2
            pic[m] = pic[m] * 0.5 + torch.randint(0, 256, (3,), generator=g, device="cuda").float() * 0.5
        pic = (pic + 2.0 * torch.randn(pic.shape, generator=g, device="cuda")).clamp(0, 255)
        return torch.cat([pic, torch.full((ROWS, COLS, 1), 255.0, device="cuda")], -1).to(torch.uint8).contiguous()

    def detectors_photo():
        src = zg.Image(photo_like())
        out = zg.Image(torch.empty((ROWS, COLS), dtype=torch.uint8, device="cuda"))
        r = rate(_time_kernel(torch, lambda i: src.canny(1.4, 50, 150, out=out), n=8, warm=2), ROWS * COLS, 5 * ROWS * COLS)
        r["shen_castan_ms"] = round(_time_kernel(torch, lambda i: src.shen_castan(out=out), n=4, warm=1), 5)
        r["note"] = "ms = Image.canny(1.4, 50, 150); one frame re-read (it fits the Infinity Cache): compare with the noise legs for the hysteresis' share, not for bandwidth"
        return r

    def pyramid_build():
        # ImagePyramid.build(source, 8, 1.2, 1.6) — ORB's default — on a grey frame: seven blur + resize levels from the
        # original (sigma up to 5.5: 35 taps), one C call (zg_pyramid_build: the levels fork over internal streams under capture)
        src = zg.Image(torch.randint(0, 256, (ROWS, COLS), dtype=torch.uint8, device="cuda"))
        ms = _time_kernel(torch, lambda i: zg.ImagePyramid.build_default(src), n=6, warm=2)  # one zg_pyramid_build per pyramid, graph-replayed like every leg
        r = rate(ms, ROWS * COLS, 2 * ROWS * COLS)
        r["eager_ms"] = round(_time_kernel(torch, lambda i: zg.ImagePyramid.build_default(src), n=6, warm=2, capture=False), 5)
        return r

    def png_frame():
        # photo-like content (smooth ramps + a little noise): uniform noise would measure nothing but deflate's worst case
        yy, xx = torch.meshgrid(torch.arange(ROWS, device="cuda"), torch.arange(COLS, device="cuda"), indexing="ij")
        smooth = torch.stack([(xx // 8) % 256, (yy // 8) % 256, ((xx + yy) // 16) % 256, torch.full_like(xx, 255)], -1)
        return (smooth + torch.randint(0, 4, (ROWS, COLS, 4), device="cuda")).clamp(0, 255).to(torch.uint8)

    def png_filter(mode):
        # the device half of png.encode: filter costs + adaptive selection + row filtering (mode -1) or one fixed filter
        src = zg.Image(png_frame())
        ms = _time_kernel(torch, lambda i: zg.png.filter_scanlines(src, mode), n=10, warm=2, capture=False)
        return rate(ms, ROWS * COLS, 8 * ROWS * COLS)  # 4 B read + 4 B (+ 1 per row) written per pixel

    def png_files():
        # whole files through the C ABI, wall clock: the host half (zlib inflate / deflate, de-filtering) dominates by design
        import time as _t
        src = zg.Image(png_frame())
        t0 = _t.perf_counter(); data = zg.png.encode(src); t_enc = _t.perf_counter() - t0
        best = 1e9
        for _ in range(3):
            t0 = _t.perf_counter(); zg.png.load_from_bytes(data); torch.cuda.synchronize(); best = min(best, _t.perf_counter() - t0)
        return {"decode_ms": round(best * 1e3, 1), "decode_Mpixels/s": round(ROWS * COLS / best / 1e6, 1), "encode_ms": round(t_enc * 1e3, 1),
                "encode_Mpixels/s": round(ROWS * COLS / t_enc / 1e6, 1), "file_MiB": round(len(data) / 2**20, 1),
                "note": "host-bound: inflate + de-filter on one core (decode), deflate level 5 in 1 MiB pieces on up to 16 host threads (encode); device share < 1 ms"}

    def jpeg_files():
        # a photo-like 4096 x 4096 frame written by Pillow (4:2:0, quality 90), decoded through the C ABI: wall clock of the
        # whole call (host Huffman decode, upload, device IDCT + chroma + colour), one caller and 16 callers on 16 streams
        import io as _io
        import time as _t
        from concurrent.futures import ThreadPoolExecutor
        try:
            from PIL import Image as _PI
        except ImportError:
            return {"skipped": "Pillow is needed to write the test file"}
        yy, xx = np.mgrid[0:ROWS, 0:COLS].astype(np.float32)
        pic = np.stack([128 + 100 * np.sin(xx / 170) * np.cos(yy / 230), 128 + 90 * np.cos(xx / 110) * np.sin(yy / 130), 128 + 110 * np.sin((xx + yy) / 290)], -1)
        pic = np.clip(pic + np.random.default_rng(0).normal(0, 4, pic.shape), 0, 255).astype(np.uint8)
        buf = _io.BytesIO()
        _PI.fromarray(pic).save(buf, "JPEG", quality=90, subsampling=2)
        data = buf.getvalue()
        zg.jpeg.load_from_bytes(data)
        best = 1e9
        for _ in range(3):
            t0 = _t.perf_counter(); zg.jpeg.load_from_bytes(data); torch.cuda.synchronize(); best = min(best, _t.perf_counter() - t0)

        def one(_):
            with torch.cuda.stream(torch.cuda.Stream()):
                zg.jpeg.load_from_bytes(data)
                torch.cuda.current_stream().synchronize()
        with ThreadPoolExecutor(16) as ex:
            list(ex.map(one, range(16)))
            t0 = _t.perf_counter(); list(ex.map(one, range(32))); par = _t.perf_counter() - t0
        dev_pic = zg.Image(torch.from_numpy(pic).cuda())
        zg.jpeg.encode(dev_pic)
        enc = 1e9
        for _ in range(2):
            t0 = _t.perf_counter(); ours = zg.jpeg.encode(dev_pic); enc = min(enc, _t.perf_counter() - t0)
        return {"decode_ms": round(best * 1e3, 1), "decode_Mpixels/s": round(ROWS * COLS / best / 1e6, 1),
                "decode_16_threads_Mpixels/s": round(32 * ROWS * COLS / par / 1e6, 1), "file_MiB": round(len(data) / 2**20, 1),
                "encode_ms": round(enc * 1e3, 1), "encode_Mpixels/s": round(ROWS * COLS / enc / 1e6, 1), "encoded_MiB": round(len(ours) / 2**20, 1),
                "note": "host-bound: Huffman decoding is one serial chain, coding runs in bands of MCU rows on up to 16 host threads; device share ~0.12 ms (IDCT 3 planes + render) / ~0.2 ms (forward DCT)"}

    def box(kind, radius=2, sharpen=False):
        ring = 8
        shape = (ROWS, COLS, 4) if kind == "rgba" else (ROWS, COLS)
        im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in u8_frames(ring, shape)]
        if sharpen:
            ms = _time_kernel(torch, lambda i: im[i % ring][0].sharpen(radius, out=im[i % ring][1]), n=16, warm=3)
        else:
            ms = _time_kernel(torch, lambda i: im[i % ring][0].box_blur(radius, out=im[i % ring][1]), n=16, warm=3)
        bpp = 8 if kind == "rgba" else 2
        return rate(ms, ROWS * COLS, bpp * ROWS * COLS)

    def conv3x3(kind):
        ring = 4
        if kind == "u8":
            srcs, bpp = u8_frames(ring, (ROWS, COLS, 4)), 8
        else:
            srcs, bpp = [torch.rand((ROWS, COLS, 4), dtype=torch.float32, device="cuda") for _ in range(ring)], 32
        im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in srcs]
        k = np.full((3, 3), 1.0 / 9.0, np.float32)
        ms = _time_kernel(torch, lambda i: im[i % ring][0].convolve(k, zg.BorderMode.replicate, out=im[i % ring][1]), n=16, warm=3)
        return rate(ms, ROWS * COLS, bpp * ROWS * COLS)

    def fused_batch16():
        nb = 16
        srcs = [torch.randint(0, 256, (nb, ROWS, COLS, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        outs = [torch.empty((nb, 1024, 1024, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
        pipe = zg.Pipeline([zg.Step.resize(1024, 1024), zg.Step.convert(zg.CS_OKLAB)])
        ms = _time_kernel(torch, lambda i: pipe.run(srcs[i % 2], out=outs[i % 2]), n=8, warm=2) / nb
        r = rate(ms, ROWS * COLS, 28 * 1024 * 1024)
        r["GB/s_sector_basis"] = round((ROWS * COLS * 4 // 2 + 12 * 1024 * 1024) / ms / 1e6, 1)
        r["frames_per_launch"] = nb
        return r

    def recipe_example():
        """The recipe of the CLI's own help text (src/cli/pipeline.zig:58-67): resize --width 800 with .lanczos, blur gaussian sigma 2, edges sobel,
        over 64 frames of 1080p in one zg_batch_pipeline call."""
        nb, rows, cols = 64, 1080, 1920
        srcs = [torch.randint(0, 256, (nb, rows, cols, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        outs = [torch.empty((nb, 450, 800, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
        pipe = zg.Pipeline([zg.Step.resize(450, 800, I.lanczos), zg.Step.gaussian_blur(2.0), zg.Step.edges_sobel()])
        ms = _time_kernel(torch, lambda i: pipe.run(srcs[i % 2], out=outs[i % 2]), n=6, warm=2)
        r = rate(ms, nb * rows * cols, nb * (rows * cols * 4 + 450 * 800 * 4))  # bound: every source frame read once, every result written once
        r["frames"] = nb
        return r

    def conv5x5():
        ring = 8
        k = np.full((5, 5), 1.0 / 25.0, np.float32)
        k[2, 2] += 0.25
        k[0, 0] -= 0.25
        im = [(zg.Image(s), zg.Image(torch.empty_like(s))) for s in u8_frames(ring, (ROWS, COLS, 4))]
        ms = _time_kernel(torch, lambda i: im[i % ring][0].convolve(k, out=im[i % ring][1]))
        return rate(ms, ROWS * COLS, 8 * ROWS * COLS)

    def traffic_ratio(r, op, bytes_alg):
        """counted HBM bytes of ONE call (every kernel of it) over its algorithmic bytes: how much of the traffic is waste"""
        if isinstance(r, dict) and "error" not in r and not os.environ.get("ZG_BENCH_NO_LIVE_TRAFFIC"):
            t = live_traffic(op, None)
            if t is not None:
                r["hbm_bytes_per_call"] = t
                r["traffic_ratio"] = round(t / bytes_alg, 3)
        return r

    def host_layer(op):
        """The layer a zignal caller gets by default (image.zig:954-994 takes host slices): zg_gaussian_blur_host / zg_resize_host on numpy arrays,
        PCIe both ways inside the call. Pageable and pinned (zg_malloc_host) memory; banded (upload, kernel and download of neighbouring bands
        overlap on three streams) and whole-frame (ZIGNAL_HIP_NO_BANDS). Best of five calls each; GB/s is per direction."""
        import ctypes as C
        L = zg.lib()
        if op == "blur":
            in_shape, out_shape, dt = (ROWS, COLS, 4), (ROWS, COLS, 4), np.float32
        else:
            in_shape, out_shape, dt = (ROWS, COLS, 4), (1024, 1024, 4), np.uint8
        nin, nout = int(np.prod(in_shape)) * np.dtype(dt).itemsize, int(np.prod(out_shape)) * np.dtype(dt).itemsize
        rng = np.random.default_rng(3)
        pins = []

        def alloc(shape, pinned):
            nb = int(np.prod(shape)) * np.dtype(dt).itemsize
            if not pinned:
                return np.empty(shape, dt)
            ptr = C.c_void_p()
            if L.zg_malloc_host(C.byref(ptr), nb) != 0:
                raise MemoryError(f"zg_malloc_host({nb})")
            pins.append(ptr)
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), (nb,)).view(dt).reshape(shape)

        out = {"bytes_up": nin, "bytes_down": nout}
        try:
            for mem in ("pageable", "pinned"):
                src, dst = alloc(in_shape, mem == "pinned"), alloc(out_shape, mem == "pinned")
                src[...] = rng.random(in_shape, dtype=np.float32) if dt == np.float32 else rng.integers(0, 256, in_shape, dtype=np.uint8)
                a, b = zg.Image(src), zg.Image(dst)
                call = (lambda: a.gaussian_blur(SIGMA, out=b)) if op == "blur" else (lambda: a.resize(b, I.bilinear))
                for mode in ("banded", "whole_frame"):
                    if mode == "whole_frame":
                        os.environ["ZIGNAL_HIP_NO_BANDS"] = "1"
                    try:
                        call()
                        best = float("inf")
                        for _ in range(5):
                            t0 = time.perf_counter()
                            call()
                            best = min(best, time.perf_counter() - t0)
                    finally:
                        os.environ.pop("ZIGNAL_HIP_NO_BANDS", None)
                    out[f"{mem}_{mode}"] = {"ms": round(best * 1e3, 3), "Mpixels/s": round(ROWS * COLS / best / 1e6, 1), "GB/s_up": round(nin / best / 1e9, 2),
                                            "GB/s_down": round(nout / best / 1e9, 2)}
                del a, b, src, dst
        finally:
            for ptr in pins:
                L.zg_free_host(ptr)
        return out

    def config1():
        """BASELINE configs[0]: 3 x 3 box blur (radius 1) of a 256 x 256 Image(u8) — plumbing: rows * cols * 255 < 2^24, so k_box_direct sums the window
        itself and the call is one launch. Device-resident (graph-replayed) and through the host layer (numpy in, numpy out)."""
        src = zg.Image(torch.randint(0, 256, (256, 256), dtype=torch.uint8, device="cuda"))
        dst = zg.Image(torch.empty((256, 256), dtype=torch.uint8, device="cuda"))
        ms = _time_kernel(torch, lambda i: src.box_blur(1, out=dst), n=64, warm=8)
        r = rate(ms, 256 * 256, 2 * 256 * 256)
        h = np.random.default_rng(4).integers(0, 256, (256, 256), dtype=np.uint8)
        ho = np.empty_like(h)
        a, b = zg.Image(h), zg.Image(ho)
        a.box_blur(1, out=b)
        best = float("inf")
        for _ in range(20):
            t0 = time.perf_counter()
            a.box_blur(1, out=b)
            best = min(best, time.perf_counter() - t0)
        r["host_layer_ms"] = round(best * 1e3, 4)
        return r

    leg("config1_box_blur_256_u8", config1)
    leg("host_layer_gaussian_rgba_f32_4096", lambda: host_layer("blur"))
    leg("host_layer_resize_rgba_u8_4096", lambda: host_layer("resize"))
    leg("pipeline_example_recipe_64x1080p_rgba_u8", recipe_example)
    leg("s4_convolve_5x5_rgba_u8_4096", conv5x5)
    leg("s5_box_blur_r2_rgba_u8_4096", lambda: traffic_ratio(box("rgba"), "box_rgba8", 8 * ROWS * COLS))
    leg("s5_box_blur_r2_u8_4096", lambda: traffic_ratio(box("grey"), "box_u8", 2 * ROWS * COLS))
    leg("s5_box_blur_r1_u8_4096", lambda: box("grey", 1))
    leg("s5_sharpen_r2_rgba_u8_4096", lambda: box("rgba", 2, True))
    leg("s4_convolve_3x3_rgba_u8_4096", lambda: conv3x3("u8"))
    leg("s4_convolve_3x3_rgba_f32_4096", lambda: conv3x3("f32"))
    leg("config3_fused_resize_oklab_16_frames_per_launch", fused_batch16)
    leg("next_sobel_rgba_u8_4096", sobel)
    leg("next_pyramid_build_default_u8_4096", lambda: traffic_ratio(pyramid_build(), "pyramid", 2 * ROWS * COLS))
    leg("next_canny_rgba_u8_4096", canny)
    leg("next_shen_castan_rgba_u8_4096", shen)
    leg("next_canny_and_shen_castan_photo_like_rgba_u8_4096", detectors_photo)
    leg("next_pyramid_level3_blur_u8_4096", pyramid_blur)
    leg("next_convert_rgba_u8_to_lab_f32_4096", lambda: lab(True))
    leg("next_convert_lab_f32_to_rgba_u8_4096", lambda: lab(False))
    leg("io_png_filter_adaptive_rgba_u8_4096", lambda: png_filter(-1))
    leg("io_png_filter_paeth_rgba_u8_4096", lambda: png_filter(4))
    leg("io_png_file_rgba_u8_4096", png_files)
    leg("io_jpeg_file_420_q90_4096", jpeg_files)
    leg("config2a_gaussian_blur_one_f32_plane_4096", blur_planes)
    leg("config2a_gaussian_blur_four_f32_planes_one_launch_4096", blur_planes4)
    leg("config2b_gaussian_blur_rgba_u8_4096", blur_u8)
    leg("config3_resize_bilinear_rgba_u8_4096_to_1024", resize_u8)
    leg("resize_bilinear_rgba_u8_8192_to_4096", lambda: resize_dense(8192, 4096))
    leg("resize_bilinear_rgba_u8_2048_to_4096", lambda: resize_dense(2048, 4096))
    leg("config3_convert_rgba_u8_to_oklab_f32_4096", oklab)
    leg("config3_fused_resize_bilinear_to_oklab_4096_to_1024", resize_oklab_fused)
    leg("config4_warp_projective_bicubic_rgba_u8_4096", lambda: warp("u8"))
    leg("config4_warp_projective_bicubic_rgba_f32_4096", lambda: warp("f32"))
    leg("config5_batch_blur_resize_64x1080p_rgba_u8", batch)
    leg("config5_batch_blur_resize_128x1080p_rgba_u8", lambda: batch(128))  # the per-GPU shard of BASELINE configs[4]: 1024 frames / 8 GPUs
    return out


if __name__ == "__main__":
    main()
