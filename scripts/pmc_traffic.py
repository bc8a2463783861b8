#!/usr/bin/env python3
"""Summarises a scripts/profile_bench.sh output directory into one JSON file.

HBM traffic of the segment kernel per launch from the FETCH_SIZE / WRITE_SIZE passes (both in
KiB; on gfx950 FETCH_SIZE tallies 128-byte read requests as 64 bytes, so reads are doubled, which
is an upper bound for the 8- and 4-byte-per-lane streams of the path pool), the kernel's average
duration from the kernel trace, and lane utilisation / wait shares from the SQ pass.
usage: pmc_traffic.py gpurun_out/prof_<tag> out.json [resident waves per SIMD of the kernel, default 3]
"""
import collections
import csv
import glob
import json
import sys


def counters(pattern):
    files = glob.glob(pattern)
    if not files:
        return {}, {}
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


def kernel_sources_sha256():
    """Digest of the library's sources: a bench line that quotes these counters says whether the kernels have changed since."""
    import hashlib
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha256()
    for f in ("mrt_device.hip", "mrt_post.hip", "mrt_host.cpp", "mrt_octree.cpp", "mrt_internal.h"):
        h.update(open(os.path.join(root, "ray_tracing_amd", "csrc", f), "rb").read())
    h.update(open(os.path.join(root, "Makefile"), "rb").read())
    return h.hexdigest()


def main():
    root, out = sys.argv[1], sys.argv[2]
    res = {"source": root, "kernel": None, "kernel_sources_sha256": kernel_sources_sha256()}
    stats = glob.glob(f"{root}/trace/*/*_kernel_stats.csv")
    if stats:
        for r in csv.DictReader(open(stats[0])):
            if "k_segments" in r["Name"]:
                res["kernel"] = r["Name"].split("(")[0]
                res["trace"] = {"calls": int(r["Calls"]), "avg_ms": float(r["AverageNs"]) / 1e6,
                                "total_ms": float(r["TotalDurationNs"]) / 1e6, "percent_of_gpu_time": float(r["Percentage"])}
    for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        agg, n = counters(f"{root}/pmc_{name}/*/*_counter_collection.csv")
        for k in agg:
            if "k_segments" in k:
                res[ctr] = {"launches": n[k], "kib_per_launch": agg[k][ctr] / n[k]}
    if "FETCH_SIZE" in res and "WRITE_SIZE" in res:
        f, w = res["FETCH_SIZE"]["kib_per_launch"], res["WRITE_SIZE"]["kib_per_launch"]
        res["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
        res["hbm_bytes_per_launch_uncorrected"] = (f + w) * 1024.0
    agg, n = counters(f"{root}/pmc_sq/*/*_counter_collection.csv")
    for k, a in agg.items():
        if "k_segments" in k and a.get("SQ_WAVE_CYCLES"):
            wc = a["SQ_WAVE_CYCLES"]
            res["sq"] = {"lane_utilisation": a["SQ_THREAD_CYCLES_VALU"] / (64.0 * a["SQ_ACTIVE_INST_VALU"]),
                         "valu_active_per_wave_cycle": a["SQ_ACTIVE_INST_VALU"] / wc,
                         "scalar_active_per_wave_cycle": a.get("SQ_ACTIVE_INST_SCA", 0.0) / wc,
                         "wait_any_per_wave_cycle": a["SQ_WAIT_ANY"] / wc,
                         "wait_inst_any_per_wave_cycle": a["SQ_WAIT_INST_ANY"] / wc}
    # f64 instruction mix (wave-level instruction counts) -> f64 flops per path segment. A wave instruction does
    # one operation on each ACTIVE lane (two for an fma); the active-lane share is the kernel's overall lane
    # utilisation from the SQ pass (the counters do not give it per instruction class).
    agg, n = counters(f"{root}/pmc_mix/*/*_counter_collection.csv")
    mix_line = None
    for f in glob.glob(f"{root}/bench_mix.json"):
        for line in open(f):
            if line.startswith("{"):
                mix_line = json.loads(line)
    for k, a in agg.items():
        if "k_segments" in k and a.get("SQ_INSTS_VALU"):
            f64_wave_ops = a.get("SQ_INSTS_VALU_ADD_F64", 0.0) + a.get("SQ_INSTS_VALU_MUL_F64", 0.0) + \
                a.get("SQ_INSTS_VALU_TRANS_F64", 0.0) + 2.0 * a.get("SQ_INSTS_VALU_FMA_F64", 0.0)
            res["f64_mix"] = {"valu_insts": a["SQ_INSTS_VALU"], "salu_insts": a.get("SQ_INSTS_SALU", 0.0),
                              "smem_insts": a.get("SQ_INSTS_SMEM", 0.0), "waves": a.get("SQ_WAVES", 0.0),
                              "add_f64": a.get("SQ_INSTS_VALU_ADD_F64", 0.0), "mul_f64": a.get("SQ_INSTS_VALU_MUL_F64", 0.0),
                              "fma_f64": a.get("SQ_INSTS_VALU_FMA_F64", 0.0), "trans_f64": a.get("SQ_INSTS_VALU_TRANS_F64", 0.0),
                              "f64_wave_flops": f64_wave_ops}
            if mix_line and "sq" in res:
                rays = mix_line["mrays_per_s"] * 1e6 * mix_line["ms_per_step"] * 1e-3 * (mix_line["steps"] + mix_line["warmup"])
                res["f64_mix"]["segments_in_run"] = rays
                res["f64_mix"]["valu_wave_insts_per_64_segments"] = a["SQ_INSTS_VALU"] * 64.0 / rays
                res["f64_mix"]["valu_lane_insts_per_segment"] = a["SQ_INSTS_VALU"] * 64.0 * res["sq"]["lane_utilisation"] / rays
                res["f64_mix"]["f64_flops_per_segment"] = f64_wave_ops * 64.0 * res["sq"]["lane_utilisation"] / rays
                res["f64_mix"]["source"] = ("rocprofv3 SQ_INSTS_VALU_{ADD,MUL,TRANS,FMA x2}_F64 wave instructions x 64 lanes x "
                                            "lane utilisation / path segments of the profiled run")
    if "sq" in res:
        waves = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
        res["sq"]["waves_per_simd"] = waves
        res["sq"]["valu_issue_busy"] = min(1.0, res["sq"]["valu_active_per_wave_cycle"] * waves)
    for f in glob.glob(f"{root}/bench_trace.json"):
        for line in open(f):
            if line.startswith("{"):
                res["bench_line_under_trace"] = json.loads(line)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
