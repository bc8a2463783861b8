#!/usr/bin/env python3
"""Benchmark: Cornell smoke 1920x1080 path tracing throughput (Mpaths/s) on N MI355X.

One step = one full render of the frame at --spp samples per pixel through the C ABI
(`mrt_render_accumulate_device`), accumulation buffer resident in HBM. With N > 1 GPUs each
rank renders --spp samples with its own disjoint sample-index range (weak scaling: the joined
image holds N x spp samples) and the ranks' buffers are summed by one all-reduce (RCCL) inside
the timed region. Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

B_SEG = 220.0   # algorithmic HBM bytes per path segment: ray 48 + id 4 read, hit 12 written; ray + hit + state 80 read,
                # next ray + state 68 written; 8 for the compaction index
B_PATH = 24.0   # algorithmic HBM bytes per path (accumulation read-modify-write)
HBM_PEAK_GBS = 8000.0


def cpu_baseline(info, width, height, spp, threads):
    """Times the CPU checker (a C++ port of the reference's CPU path) on a bounded sample."""
    from tests.oracle_ffi import OracleWorld

    ow = OracleWorld(info)
    t0 = time.perf_counter()
    _, st = ow.render(width, height, spp, seed=0, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": st.paths / dt / 1e6, "unit": "Mpaths/s", "cores": threads, "kind": "port",
            "mrays_per_s": st.rays / dt / 1e6, "seconds": dt,
            "sample": f"Cornell smoke {width}x{height} @ {spp} spp, {threads} threads over vertical strips "
                      f"(threaded_render layout)"}


def pmc_traffic(workload, launches_per_step):
    """HBM bytes per segment-kernel launch from the committed rocprofv3 counter passes
    (profiles/*_pmc_traffic.json, written by scripts/pmc_traffic.py), or None when no pass of this
    workload with the same launch count per step is on file. Counters cannot be read from inside
    the timed run, so the value is that of the profiled run of this same command."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
            line = d.get("bench_line_under_trace", {})
            same = line.get("config", {}).get("workload") == workload
            per_step = line.get("roofline", {}).get("launches", 0) / max(1, line.get("steps", 1))
            if same and abs(per_step - launches_per_step) <= 1 and "hbm_bytes_per_launch" in d:
                best = float(d["hbm_bytes_per_launch"])
        except (OSError, ValueError):
            continue
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=1024)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--pool-slots", type=int, default=0)
    ap.add_argument("--segments-per-launch", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-spp", type=int, default=64)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from ray_tracing_amd import RayTracer, scenarios
    from ray_tracing_amd.distributed import render_sharded

    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world_size != args.gpus and world_size > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world_size}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    if world_size > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    info = scenarios.cornell_smoke()
    rt = RayTracer(info, device=local_rank, seed=0)
    if args.pool_slots:
        rt.world.set_option("pool_slots", args.pool_slots)
    if args.segments_per_launch:
        rt.world.set_option("segments_per_launch", args.segments_per_launch)
    W, H, spp = args.width, args.height, args.spp
    accum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream()
    totals = {"paths": 0, "rays": 0, "lq": 0, "kernel_ms": 0.0, "launches": 0}

    def render(first, count):
        st = rt.render_device(accum.data_ptr(), W, H, count, first_sample=first, stream=stream.cuda_stream)
        totals["paths"] += st.paths
        totals["rays"] += st.rays
        totals["lq"] += st.light_queries
        totals["kernel_ms"] += st.kernel_ms
        totals["launches"] += st.launches

    def step():
        accum.zero_()
        # weak scaling: every rank renders `spp` samples, sample indices [rank*spp, (rank+1)*spp)
        render_sharded(render, lambda: dist.all_reduce(accum), spp * world_size, rank, world_size)

    def sync():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    for k in totals:
        totals[k] = 0 if k != "kernel_ms" else 0.0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    sync()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([totals["paths"], totals["rays"], totals["lq"]], dtype=torch.float64, device="cuda")
        dist.all_reduce(c)
        all_paths, all_rays, all_lq = (float(x) for x in c.tolist())
    else:
        all_paths, all_rays, all_lq = float(totals["paths"]), float(totals["rays"]), float(totals["lq"])

    if rank == 0:
        mean = accum.mean().item() / (spp * world_size)
        kernel_s = totals["kernel_ms"] / 1e3
        algo_bytes = B_SEG * totals["rays"] + B_PATH * totals["paths"]   # this rank's segment kernels
        achieved = algo_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        workload = f"cornell_smoke {W}x{H} @ {spp} spp per GPU, depth 50, seed 0"
        out = {
            "metric": "Mpaths/s, Cornell smoke 1920x1080",
            "value": all_paths / elapsed / 1e6,
            "unit": "Mpaths/s",
            "n_gpus": world_size,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": workload,
                       "parallelism": f"spp-split x{world_size} + one all-reduce of the f32 accumulation buffer"},
            "mrays_per_s": all_rays / elapsed / 1e6,
            "light_queries_per_s_M": all_lq / elapsed / 1e6,
            "segments_per_path": all_rays / all_paths if all_paths else 0.0,
            "mean_radiance": mean,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic(workload, totals["launches"] / max(1, args.steps)),
                         "kernel": "k_segments", "avg_launch_ms": totals["kernel_ms"] / max(1, totals["launches"]),
                         "launches": totals["launches"],
                         "algorithmic_bytes_per_launch": algo_bytes / max(1, totals["launches"])},
        }
        if world_size == 1 and not args.no_cpu_baseline:
            threads = min(16, os.cpu_count() or 1)
            cb = cpu_baseline(info, W, H, args.cpu_spp, threads)
            # the reference's own layout: eight worker threads over eight vertical strips (ray_tracer.rs:496-497)
            ref8 = cpu_baseline(info, W, H, max(1, args.cpu_spp // 8), 8)
            cb["reference_layout_8_threads"] = {k: ref8[k] for k in ("value", "unit", "cores", "seconds", "sample")}
            out["cpu_baseline"] = cb
            out["gpu_over_cpu"] = out["value"] / cb["value"] if cb["value"] else None
        print(json.dumps(out), flush=True)
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
