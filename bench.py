#!/usr/bin/env python3
"""Benchmark: path-tracing throughput (Mpaths/s) of the HIP path on N MI355X, one rank per GPU.

One step = one full render of the frame at --spp samples per pixel through the C ABI
(`mrt_render_accumulate_device`), accumulation buffer resident in HBM. Default workload: Cornell
smoke 1920x1080 at 1024 spp.

N > 1: samples are split across ranks by sample index and the ranks' buffers are summed by ONE
all-reduce (RCCL through torch.distributed's "nccl" backend) inside the timed region.
  weak scaling (default): every rank renders --spp samples, the joined image holds N x spp samples;
  --strong:               --spp is the total, rank r renders its share of the sample indices
                          (3840x2160 at 16384 spp over eight GPUs is `--strong --width 3840
                          --height 2160 --spp 16384 --gpus 8`).
Started without a launcher (`python bench.py --gpus N`, WORLD_SIZE unset) the parent process
starts the N ranks itself as fresh child processes, before it touches torch or the GPU, and passes
rank 0's JSON line through. Started under torch.distributed.run it is one of the ranks.
`--backend gloo --dry` rehearses the rank plumbing on CPUs (no rendering, no GPU).
"""
from __future__ import annotations

import argparse
import datetime
import glob
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_SEG = 220.0   # algorithmic HBM bytes per path segment of a queue-based wavefront tracer: ray 48 + id 4 read,
                # hit 12 written; ray + hit + state 80 read, next ray + state 68 written; 8 for the compaction index
B_PATH = 24.0   # algorithmic HBM bytes per path (accumulation read-modify-write)
HBM_PEAK_GBS = 8000.0
FP64_VECTOR_PEAK_TFLOPS = 78.6


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="Cornell Smoke", help="scenario name (ray_tracing_amd.scenarios) or 'Voxel City'")
    ap.add_argument("--strong", action="store_true", help="--spp is the total over all ranks (strong scaling)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--dry", action="store_true", help="no rendering and no GPU: rank plumbing and all-reduce only")
    ap.add_argument("--pool-slots", type=int, default=0)
    ap.add_argument("--segments-per-launch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=64, help="samples per pixel of the CPU baseline's sample of the workload")
    ap.add_argument("--cpu-ref8-spp", type=int, default=0,
                    help="samples per pixel of the eight-thread (reference layout) CPU leg; default cpu-spp / 8")
    ap.add_argument("--post", action="store_true", help="also time bloom + to_image on the device-resident image")
    return ap.parse_args(argv)


def launch_ranks(args) -> int:
    """Starts args.gpus fresh rank processes of this script (nothing in this process has touched
    torch or the GPU), watches ALL of them and passes rank 0's output through. When any rank ends with
    an error the others are ended within seconds (they would otherwise sit in the rendezvous or in a
    collective until its timeout) and the error code is returned. Every rank's stderr is kept in
    gpurun_out/rank<k>.err."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    log_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(log_dir, exist_ok=True)
    procs, errs = [], []
    out0 = open(os.path.join(log_dir, "rank0.out"), "w+")
    for rank in range(args.gpus):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        err = open(os.path.join(log_dir, f"rank{rank}.err"), "w+")
        errs.append(err)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if rank == 0 else subprocess.DEVNULL, stderr=err))
    failed = None
    while failed is None and any(p.poll() is None for p in procs):
        for rank, p in enumerate(procs):
            if p.poll() not in (None, 0):
                failed = rank
                break
        else:
            time.sleep(0.1)
    if failed is None:
        failed = next((rank for rank, p in enumerate(procs) if p.returncode != 0), None)
    if failed is not None:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        deadline = time.monotonic() + 5.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        errs[failed].seek(0)
        sys.stderr.write(f"bench.py: rank {failed} ended with code {procs[failed].returncode}; the other ranks were stopped. "
                         f"Its stderr (gpurun_out/rank{failed}.err) ends:\n{errs[failed].read()[-2000:]}\n")
    out0.seek(0)
    sys.stdout.write(out0.read())
    sys.stdout.flush()
    for f in errs + [out0]:
        f.close()
    if failed is not None:
        return abs(procs[failed].returncode) or 1
    return 0


def cpu_baseline(info, width, height, spp, threads, label):
    """Times the CPU checker (a C++ port of the reference's CPU path) on a bounded sample, in sample-range
    chunks with a progress line on stderr (a slow scene would otherwise look hung)."""
    import numpy as np
    from tests.oracle_ffi import OracleWorld

    ow = OracleWorld(info)
    accum = np.zeros((height, width, 3), dtype=np.float32)
    paths = rays = 0
    t0 = time.perf_counter()
    done = 0
    chunk = max(1, spp // 8)
    while done < spp:
        n = min(chunk, spp - done)
        _, st = ow.render(width, height, n, seed=0, first_sample=done, threads=threads, accum=accum)
        paths, rays, done = paths + st.paths, rays + st.rays, done + n
        print(f"[cpu baseline] {label}: {done}/{spp} spp on {threads} threads, {time.perf_counter() - t0:.1f} s",
              file=sys.stderr, flush=True)
    dt = time.perf_counter() - t0
    return {"value": paths / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "mrays_per_s": rays / dt / 1e6, "seconds": dt,
            "sample": f"{label} {width}x{height} @ {spp} spp, {threads} threads over vertical strips "
                      f"(threaded_render layout)"}


def native_checker_build():
    """Builds the CPU checker once more on this host with -march=native (the prebuilt library is compiled
    off the GPU box for a generic x86-64) and points the binding at it. Returns the description of the build
    that will be timed. Falls back to the prebuilt library when there is no compiler or the build fails."""
    import shutil
    import subprocess
    import tempfile
    from tests import oracle_ffi

    generic = "g++ -O3 -ffp-contract=off, generic x86-64 (prebuilt off the GPU box)"
    cxx = shutil.which("g++")
    if cxx is None or oracle_ffi._lib is not None:
        return generic
    out_dir = tempfile.mkdtemp(prefix="mrt_checker_")
    lib = os.path.join(out_dir, "libmrt_oracle_native.so")
    srcs = [os.path.join(ROOT, "oracle", f) for f in ("cpu_ref.cpp", "octree_api.cpp", "post_ref.cpp")]
    cmd = [cxx, "-O3", "-march=native", "-std=c++17", "-fPIC", "-ffp-contract=off", "-shared", "-pthread",
           "-I" + os.path.join(ROOT, "include"), "-o", lib] + srcs
    try:
        subprocess.run(cmd, check=True, timeout=180, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    except (OSError, subprocess.SubprocessError):
        return generic
    prebuilt = oracle_ffi.ORACLE_PATH
    oracle_ffi.ORACLE_PATH = lib
    try:
        oracle_ffi.load()
    except (OSError, ImportError):
        oracle_ffi.ORACLE_PATH = prebuilt
        return generic
    return "g++ -O3 -march=native -ffp-contract=off, built on this host for the timing"


def host_description():
    """The host CPU as /proc/cpuinfo names it. A virtualised host may show only the family name under
    "model name", so the numeric family / model / stepping, which identify the part, are kept too."""
    fields = {}
    try:
        for line in open("/proc/cpuinfo"):
            if ":" not in line:
                if fields:
                    break          # end of the first processor's block
                continue
            k, v = (x.strip() for x in line.split(":", 1))
            if k in ("model name", "vendor_id", "cpu family", "model", "stepping", "cache size", "cpu MHz"):
                fields.setdefault(k, v)
    except OSError:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    model = fields.get("model name", "unknown")
    ident = " ".join(f"{k} {fields[k]}" for k in ("vendor_id", "cpu family", "model", "stepping") if k in fields)
    return {"cpu_model": f"{model} ({ident}; cache {fields.get('cache size', '?')}; {fields.get('cpu MHz', '?')} MHz)",
            "nproc": os.cpu_count() or 1, "usable_cores": usable}


def committed_profile(workload_key):
    """Counter figures of the segment kernel from the committed rocprofv3 passes of this workload
    (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.py from scripts/profile_bench.sh
    runs of this same command; counters cannot be read inside the timed run). Newest file wins."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        line = d.get("bench_line_under_trace", {})
        if line.get("config", {}).get("profile_key") == workload_key or \
                (workload_key.startswith("cornell_smoke 1920x1080 @ 1024") and
                 str(line.get("config", {}).get("workload", "")).startswith("cornell_smoke 1920x1080 @ 1024")):
            d["file"] = os.path.relpath(f, ROOT)
            best = d
    return best


def scene_name(name):
    """Display name of a scenario given either the display name or its key form (`random_scene`, `voxel_city`)."""
    from ray_tracing_amd import scenarios
    names = list(scenarios.SCENARIOS) + list(scenarios.ASSET_SCENARIOS)
    key = name.lower().replace(",", "").replace(" ", "_")
    for n in names:
        if n.lower().replace(",", "").replace(" ", "_") == key:
            return n
    raise SystemExit(f"unknown scene {name!r}; known: {sorted(names)}")


def build_scene(name):
    from ray_tracing_amd import scenarios
    # the decoded asset files (.vox models, tile map, earthmap) are kept as data under tests/golden
    return scenarios.build_with_assets(name, os.path.join(ROOT, "tests", "golden"))


def post_leg(rt, accum, W, H, spp, repeats=5):
    """bloom(accum / spp) and the 8-bit mapping on the device-resident image, HIP-event timed."""
    import torch
    out = torch.empty_like(accum)
    rgb8 = torch.empty((H, W, 3), dtype=torch.uint8, device=accum.device)
    stream = torch.cuda.current_stream().cuda_stream
    res = {}
    for name, fn in (("bloom", lambda: rt.bloom_device(accum.data_ptr(), W, H, spp, out.data_ptr(), stream)),
                     ("to_image", lambda: rt.to_image_device(out.data_ptr(), W, H, 1, rgb8.data_ptr(), stream))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(repeats):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_ms"] = e0.elapsed_time(e1) / repeats
    img_bytes = 12.0 * W * H
    # bloom: select 1 read + 2 writes; per level (sizes 1, 1/4, ...): down-sample, blur x, blur y (read + write each),
    # up-add (2 reads + 1 write); tone map 2 reads + 1 write -> about 3 + 1.333 * (2 + 2 + 2 + 3) + 3 = 18 image passes
    res["bloom_algorithmic_bytes"] = 18.0 * img_bytes
    res["bloom_gbs"] = res["bloom_algorithmic_bytes"] / (res["bloom_ms"] * 1e-3) / 1e9
    res["bloom_hbm_frac"] = res["bloom_gbs"] / HBM_PEAK_GBS
    res["to_image_gbs"] = (img_bytes + 3.0 * W * H) / (res["to_image_ms"] * 1e-3) / 1e9
    res["image"] = f"{W}x{H}"
    return res


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist

    from ray_tracing_amd.distributed import render_sharded, sample_range

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    use_gpu = not args.dry
    if use_gpu:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback); --dry rehearses the ranks on CPUs")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"rank {rank}: no GPU {local_rank} on this node")
        torch.cuda.set_device(local_rank)
    if os.environ.get("MRT_BENCH_FAIL_RANK") == str(rank):    # rehearsal of a rank that dies before the rendezvous
        raise SystemExit(f"rank {rank}: failure injected by MRT_BENCH_FAIL_RANK")
    if world_size > 1:
        # a rank that never arrives must not hold the others for torch's default half hour
        limit = datetime.timedelta(seconds=float(os.environ.get("MRT_BENCH_RENDEZVOUS_S", "300")))
        if args.backend == "nccl" and use_gpu:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=limit)
        else:
            dist.init_process_group("gloo", timeout=limit)

    args.scene = scene_name(args.scene)
    W, H = args.width, args.height
    total_spp = args.spp if args.strong else args.spp * world_size
    my_first, my_spp = sample_range(total_spp, rank, world_size)
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
    totals = {"paths": 0, "rays": 0, "lq": 0, "kernel_ms": 0.0, "launches": 0}

    rt = info = None
    if use_gpu:
        from ray_tracing_amd import RayTracer
        info = build_scene(args.scene)
        rt = RayTracer(info, device=local_rank, seed=0)
        if args.pool_slots:
            rt.world.set_option("pool_slots", args.pool_slots)
        if args.segments_per_launch:
            rt.world.set_option("segments_per_launch", args.segments_per_launch)
        stream = torch.cuda.current_stream()

    def render(first, count):
        if not use_gpu:
            accum.add_(float(count))          # stands in for `count` samples of radiance 1
            totals["paths"] += W * H * count
            return
        st = rt.render_device(accum.data_ptr(), W, H, count, first_sample=first, stream=stream.cuda_stream)
        totals["paths"] += st.paths
        totals["rays"] += st.rays
        totals["lq"] += st.light_queries
        totals["kernel_ms"] += st.kernel_ms
        totals["launches"] += st.launches

    ar_events, ar_host_ms = [], [0.0]

    def all_reduce():
        """The one collective of a step, timed on its own: device events around it on the stream the render ran on
        (the collective's own stream is joined to it), the host clock for gloo. The time includes the wait for the
        slowest rank to arrive."""
        if use_gpu:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            dist.all_reduce(accum)
            e1.record()
            ar_events.append((e0, e1))
        else:
            t = time.perf_counter()
            dist.all_reduce(accum)
            ar_host_ms[0] += (time.perf_counter() - t) * 1e3

    def step():
        accum.zero_()
        render_sharded(render, all_reduce, total_spp, rank, world_size)

    def sync():
        if world_size > 1:
            dist.barrier()
        if use_gpu:
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for k in totals:
        totals[k] = 0 if k != "kernel_ms" else 0.0
    sync()
    ar_events.clear()
    ar_host_ms[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    collective = None
    if world_size > 1:
        own_elapsed = elapsed
        ar_ms = ar_host_ms[0] + sum(e0.elapsed_time(e1) for e0, e1 in ar_events)
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([totals["paths"], totals["rays"], totals["lq"]], dtype=torch.float64, device=device)
        dist.all_reduce(c)
        all_paths, all_rays, all_lq = (float(x) for x in c.tolist())
        # what every rank did, as the ranks themselves saw it: proof that the collective joined world_size ranks
        mine = torch.tensor([float(rank), float(totals["paths"]), own_elapsed, ar_ms, float(my_spp)], dtype=torch.float64,
                            device=device)
        gathered = [torch.zeros_like(mine) for _ in range(world_size)]
        dist.all_gather(gathered, mine)
        rows = sorted((g.tolist() for g in gathered), key=lambda r: r[0])
        collective = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                      "ranks_seen": [int(r[0]) for r in rows],
                      "bytes": accum.numel() * accum.element_size(), "calls_per_step": 1,
                      "allreduce_ms": ar_ms / max(1, args.steps),
                      "allreduce_ms_per_rank": [r[3] / max(1, args.steps) for r in rows],
                      "allreduce_ms_is": "per step, timed around the call on each rank: transfer + wait for the slowest rank",
                      "spp_per_rank": [int(r[4]) for r in rows]}
        per_rank_mpaths = [r[1] / r[2] / 1e6 if r[2] > 0 else 0.0 for r in rows]
    else:
        all_paths, all_rays, all_lq = float(totals["paths"]), float(totals["rays"]), float(totals["lq"])
        per_rank_mpaths = [all_paths / elapsed / 1e6]

    if rank == 0:
        mean = accum.double().mean().item() / max(1, total_spp)
        kernel_s = totals["kernel_ms"] / 1e3
        launches = max(1, totals["launches"])
        algo_bytes = B_SEG * totals["rays"] + B_PATH * totals["paths"]   # this rank's segment kernels
        achieved = algo_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        scene_key = args.scene.lower().replace(",", "").replace(" ", "_")
        per_gpu = f"{my_spp} spp per GPU" if not args.strong else f"{total_spp} spp split over {world_size} GPUs"
        workload = f"{scene_key} {W}x{H} @ {per_gpu}, depth 50, seed 0"
        profile_key = f"{scene_key} {W}x{H} @ {my_spp} spp per rank"
        out = {
            "metric": f"Mpaths/s, {args.scene} {W}x{H}" if scene_key != "cornell_smoke" or (W, H) != (1920, 1080)
                      else "Mpaths/s, Cornell smoke 1920x1080",
            "value": all_paths / elapsed / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong" if args.strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload, "profile_key": profile_key,
                       "parallelism": f"spp-split x{world_size} + one all-reduce of the f32 accumulation buffer "
                                      f"({args.backend if world_size > 1 else 'single rank'})"},
            "mrays_per_s": all_rays / elapsed / 1e6,
            "light_queries_per_s_M": all_lq / elapsed / 1e6,
            "segments_per_path": all_rays / all_paths if all_paths else 0.0,
            "mean_radiance": mean,
            "per_rank_mpaths": per_rank_mpaths,
        }
        if collective:
            out["collective"] = collective
        if args.dry:
            out["dry"] = True
            out["config"]["workload"] = "dry run: no rendering, rank plumbing and all-reduce only"
        else:
            avg_launch_s = kernel_s / launches
            roof = {
                # What limits the kernel is vector-ALU issue under divergence (see the counter figures below),
                # not HBM: path state stays in registers between the segments of a launch. `achieved` / `frac`
                # keep the ALGORITHMIC-bytes accounting (220 B per segment + 24 B per path, the traffic of a
                # queue-based wavefront tracer that writes rays and hits to memory between its stages) over the kernel's HIP-event time; `measured_hbm_frac` is
                # what the kernel really moves.
                "bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "frac_is": "algorithmic bytes / kernel time / HBM peak",
                "traffic": None, "kernel": "k_segments", "avg_launch_ms": avg_launch_s * 1e3,
                "launches": totals["launches"], "algorithmic_bytes_per_launch": algo_bytes / launches,
            }
            prof = committed_profile(profile_key)
            if prof:
                roof["counters_from"] = prof["file"]
                # the counters describe the kernels as they were when the profile was taken
                try:
                    sys.path.insert(0, os.path.join(ROOT, "scripts"))
                    from pmc_traffic import kernel_sources_sha256
                    roof["counters_match_current_sources"] = prof.get("kernel_sources_sha256") == kernel_sources_sha256()
                except (ImportError, OSError):
                    roof["counters_match_current_sources"] = None
                prof_launches_per_step = prof.get("bench_line_under_trace", {}).get("roofline", {}).get("launches", 0) / \
                    max(1, prof.get("bench_line_under_trace", {}).get("steps", 1))
                same_launches = abs(prof_launches_per_step - totals["launches"] / max(1, args.steps)) <= 1
                if "hbm_bytes_per_launch" in prof and same_launches:
                    roof["traffic"] = float(prof["hbm_bytes_per_launch"])
                    roof["measured_hbm_frac"] = roof["traffic"] / avg_launch_s / (HBM_PEAK_GBS * 1e9)
                sq = prof.get("sq", {})
                if sq:
                    roof["lane_utilisation"] = sq.get("lane_utilisation")
                    roof["valu_active_per_wave_cycle"] = sq.get("valu_active_per_wave_cycle")
                    roof["valu_issue_busy"] = sq.get("valu_issue_busy")
                mix = prof.get("f64_mix", {})
                if mix.get("f64_flops_per_segment"):
                    roof["fp64_flops_per_segment"] = mix["f64_flops_per_segment"]
                    roof["fp64_flops_source"] = mix.get("source")
                    roof["fp64_flops_frac"] = mix["f64_flops_per_segment"] * totals["rays"] / kernel_s / \
                        (FP64_VECTOR_PEAK_TFLOPS * 1e12)
            out["roofline"] = roof
            if args.post:
                out["post"] = post_leg(rt, accum, W, H, total_spp)
            if world_size == 1 and not args.no_cpu_baseline:
                host = host_description()
                threads = min(16, host["usable_cores"])
                label = args.scene
                build = native_checker_build()
                cb = cpu_baseline(info, W, H, args.cpu_spp, threads, label)
                # the reference's own layout: eight worker threads over eight vertical strips (ray_tracer.rs:496-497)
                ref8 = cpu_baseline(info, W, H, args.cpu_ref8_spp or max(1, args.cpu_spp // 8), 8, label)
                cb["reference_layout_8_threads"] = {k: ref8[k] for k in ("value", "unit", "cores", "seconds", "sample", "mrays_per_s")}
                cb["host"] = host
                cb["build"] = build
                out["cpu_baseline"] = cb
                out["gpu_over_cpu"] = out["value"] / cb["value"] if cb["value"] else None
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
