#!/usr/bin/env python3
"""Records the statistical-parity figures of the HIP path against the CPU checker (test infrastructure; run on a GPU box).

Protocol: linear pre-bloom radiance `accum / spp` (what `tracing_loop` holds before `post_process`,
ray_tracer.rs:469-477). For one scene, image size and sample count S:
  R_cc = RMSE between two checker renders with independent sequential generators (mt19937_64, different seeds),
  R_gc = mean of RMSE(GPU, checker a) and RMSE(GPU, checker b), the GPU render using the addressed generator.
The GPU image passes iff R_gc <= 1.15 * R_cc and its channel means are within 0.5 % of the checkers' (plus twice the
checkers' own difference of means, their noise). Both are evaluated on linear values and on display values
`clamp(sqrt(x))` (rgb_color.rs:75-80), which damp fireflies.

The checker walks the reference's tree the reference's way, so its speed differs by two orders of magnitude between
scenes; a row whose S = 1024 render would not fit `--cpu-seconds` per checker image takes the largest power of two
that does, and says so in the row (`spp`, `spp_wanted`).

  python tests/rmse_record.py --rows small --out profiles/r03_rmse.json
  python tests/rmse_record.py --rows full  --out profiles/r03_rmse.json      (merges into the same file)
  python tests/rmse_record.py --rows random --checker-only --cache tests/_rmse_cache --cpu-seconds 5000   (no GPU needed)
  python tests/rmse_record.py --rows random --cache tests/_rmse_cache --cpu-seconds 5000 --out profiles/r03_rmse.json
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SMALL = [("Cornell Smoke", 480, 270, 1024), ("Light Demonstration", 480, 270, 1024), ("Voxel City", 480, 270, 1024),
         ("Random Scene", 240, 135, 1024)]
FULL = [("Cornell Smoke", 1920, 1080, 1024)]
# Random Scene at the protocol's size: an hour per checker image, so its two images are rendered beforehand
# (`--checker-only --cache DIR`, any host) and the GPU box only compares (`--cache DIR`).
RANDOM = [("Random Scene", 480, 270, 1024)]


def display(x):
    return np.clip(np.sqrt(np.maximum(x, 0.0)), 0.0, 1.0)


def rmse(a, b):
    d = a.astype(np.float64) - b.astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


def checker_render(ow, W, H, spp, seed, threads, label, partial=None):
    """One checker image in sample-range chunks (a progress line per chunk). `partial`: file stem that keeps the
    accumulation after every chunk, so that an interrupted render of an hour goes on where it stopped."""
    acc = np.zeros((H, W, 3), dtype=np.float32)
    chunk = max(1, spp // 16)
    done, spent = 0, 0.0
    if partial and os.path.exists(partial + ".partial.json"):
        meta = json.load(open(partial + ".partial.json"))
        acc, done, spent = np.load(partial + ".partial.npy"), meta["done"], meta["seconds"]
        print(f"  {label}: resuming at {done}/{spp} spp", flush=True)
    t0 = time.perf_counter() - spent
    while done < spp:
        n = min(chunk, spp - done)
        ow.render(W, H, n, seed=seed, first_sample=done, threads=threads, rng_mode=1, accum=acc)
        done += n
        print(f"  {label}: {done}/{spp} spp, {time.perf_counter() - t0:.1f} s", flush=True)
        if partial:
            np.save(partial + ".partial.npy", acc)
            json.dump({"done": done, "seconds": time.perf_counter() - t0}, open(partial + ".partial.json", "w"))
    if partial:
        for ext in (".partial.npy", ".partial.json"):
            if os.path.exists(partial + ext):
                os.remove(partial + ext)
    return acc / np.float32(spp), time.perf_counter() - t0


def compare(g, a, b, f):
    ga, fa, fb = f(g), f(a), f(b)
    r_cc = rmse(fa, fb)
    r_gc = 0.5 * (rmse(ga, fa) + rmse(ga, fb))
    mg = ga.mean(axis=(0, 1), dtype=np.float64)
    ma, mb = fa.mean(axis=(0, 1), dtype=np.float64), fb.mean(axis=(0, 1), dtype=np.float64)
    mc = 0.5 * (ma + mb)
    noise = np.abs(ma - mb)
    ok_means = bool((np.abs(mg - mc) <= 0.005 * mc + 2.0 * noise + 1e-6).all())
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(mc > 0, mg / mc, 1.0)
    return {"R_cc": r_cc, "R_gc": r_gc, "R_gc_over_R_cc": r_gc / r_cc if r_cc > 0 else None,
            "pass_rmse": bool(r_gc <= 1.15 * r_cc), "channel_means_gpu": mg.tolist(), "channel_means_checker": mc.tolist(),
            "channel_mean_ratio": ratio.tolist(), "checker_mean_noise": noise.tolist(), "pass_means": ok_means}


def cached_checker(cache, ow, name, W, H, spp, seed, threads, label):
    """checker_render through a directory of finished images (`<scene>_<W>x<H>_<spp>spp_seed<seed>.npy` + `.json`)."""
    if not cache:
        a, s = checker_render(ow, W, H, spp, seed, threads, label)
        return a, s, threads
    stem = os.path.join(cache, f"{name.lower().replace(' ', '_')}_{W}x{H}_{spp}spp_seed{seed}")
    if os.path.exists(stem + ".npy"):
        meta = json.load(open(stem + ".json"))
        print(f"  {label}: from {stem}.npy ({meta['seconds']:.0f} s on {meta['threads']} threads when it was rendered)", flush=True)
        return np.load(stem + ".npy"), meta["seconds"], meta["threads"]
    os.makedirs(cache, exist_ok=True)
    a, s = checker_render(ow, W, H, spp, seed, threads, label, partial=stem)
    np.save(stem + ".npy", a)
    json.dump({"seconds": s, "threads": threads}, open(stem + ".json", "w"))
    return a, s, threads


def run_row(name, W, H, spp_wanted, cpu_seconds, threads, cache=None, checker_only=False):
    from ray_tracing_amd import scenarios
    from tests.oracle_ffi import OracleWorld

    info = scenarios.build_with_assets(name, os.path.join(ROOT, "tests", "golden"))
    ow = OracleWorld(info)
    # the checker's rate on this scene, from a short render
    t0 = time.perf_counter()
    probe_spp = 1
    ow.render(W, H, probe_spp, seed=9, threads=threads, rng_mode=1)
    dt = time.perf_counter() - t0
    rate = W * H * probe_spp / dt
    spp = spp_wanted
    while spp > 8 and W * H * spp / rate > cpu_seconds:
        spp //= 2
    print(f"{name} {W}x{H}: checker {rate / 1e6:.3f} Mpaths/s on {threads} threads -> {spp} spp (wanted {spp_wanted})", flush=True)
    if checker_only:
        cached_checker(cache, ow, name, W, H, spp, 100, threads, "checker a")
        cached_checker(cache, ow, name, W, H, spp, 200, threads, "checker b")
        return None
    from ray_tracing_amd import ParallelImage, RayTracer
    rt = RayTracer(info, device=0, seed=3)
    img = ParallelImage.new_black(W, H)
    t0 = time.perf_counter()
    st = rt.trace_part(img.split(1)[0], spp, first_sample=0)
    gpu_s = time.perf_counter() - t0
    g = img.buffer / np.float32(spp)
    a, sa, threads_a = cached_checker(cache, ow, name, W, H, spp, 100, threads, "checker a")
    b, sb, threads_b = cached_checker(cache, ow, name, W, H, spp, 200, threads, "checker b")
    threads = threads_a if threads_a == threads_b else [threads_a, threads_b]
    row = {"scene": name, "width": W, "height": H, "spp": spp, "spp_wanted": spp_wanted,
           "gpu": {"seed": 3, "generator": "addressed (include/mrt.h)", "seconds": gpu_s, "rays": int(st.rays)},
           "checker": {"seeds": [100, 200], "generator": "sequential mt19937_64", "threads": threads,
                       "seconds": [sa, sb], "mpaths_per_s": W * H * spp / (0.5 * (sa + sb)) / 1e6},
           "linear": compare(g, a, b, lambda x: x), "display": compare(g, a, b, display)}
    row["pass"] = bool(row["linear"]["pass_rmse"] and row["linear"]["pass_means"] and
                       row["display"]["pass_rmse"] and row["display"]["pass_means"])
    print(json.dumps({k: row[k] for k in ("scene", "width", "height", "spp", "pass")}),
          "linear R_gc/R_cc %.4f" % row["linear"]["R_gc_over_R_cc"], "display %.4f" % row["display"]["R_gc_over_R_cc"], flush=True)
    return row


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="small", choices=["small", "full", "all", "random"])
    ap.add_argument("--cache", default=None, help="directory of finished checker images (read, and written when missing)")
    ap.add_argument("--spp", type=int, default=0, help="samples per pixel instead of the rows' own (e.g. what the cache holds)")
    ap.add_argument("--checker-only", action="store_true", help="render the checker images into --cache and stop (no GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=150.0, help="bound of one checker image")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "rmse.json"))
    args = ap.parse_args()
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    threads = args.threads or min(16, usable)
    rows = {"small": SMALL, "full": FULL, "all": SMALL + FULL, "random": RANDOM}[args.rows]
    doc = {"protocol": "R_cc = RMSE(checker a, checker b), R_gc = mean RMSE(GPU, checker a|b) at equal spp on accum / spp; "
                       "pass iff R_gc <= 1.15 R_cc and channel means within 0.5 % + 2 x |mean a - mean b|, on linear and on "
                       "clamp(sqrt(x)) values", "rows": []}
    if os.path.exists(args.out):
        try:
            doc = json.load(open(args.out))
        except ValueError:
            pass
    for name, W, H, spp in rows:
        spp = args.spp or spp
        row = run_row(name, W, H, spp, args.cpu_seconds, threads, args.cache, args.checker_only)
        if row is None:
            continue
        doc["rows"] = [r for r in doc["rows"] if (r["scene"], r["width"], r["height"]) != (name, W, H)] + [row]
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
    print("all rows pass" if all(r["pass"] for r in doc["rows"]) else "SOME ROWS FAIL", flush=True)
    return 0 if all(r["pass"] for r in doc["rows"]) else 1


if __name__ == "__main__":
    sys.exit(main())
