#!/usr/bin/env python3
"""Measures what `test_rays_at_rectangle_borders` bounds (test infrastructure; run on a GPU box): on rays aimed at the
borders of untransformed rectangles and at the touching points of untransformed spheres, how often the device's object id
differs from the CPU checker's, how many of those are exact ties, and how often the two device walks differ.

  python tests/border_rates.py --out profiles/r03_border_rates.json
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SCENES = ["Cornell Smoke", "Easy Cornell Box", "Easy Scene", "Light Demonstration", "Two Sphere"]
TIE_RTOL = 1e-9


def main():
    from ray_tracing_amd import RayTracer
    from tests import oracle_ffi
    from tests.helpers import scene_by_name
    from tests.test_gpu_parity import rectangle_border_rays

    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "border_rates.json"))
    args = ap.parse_args()
    rows = []
    for name in SCENES:
        info = scene_by_name(name)
        rt = RayTracer(info, device=0, seed=3)
        ow = oracle_ffi.OracleWorld(info)
        rays = rectangle_border_rays(info, per_rect=3000, seed=77)
        og, tg, *_ = rt.world.trace_rays(rays, seed=9)
        rt.world.set_option("traversal", 1)
        ol, tl, *_ = rt.world.trace_rays(rays, seed=9)
        rt.world.set_option("traversal", 0)
        oo, to, *_ = ow.trace_rays(rays, seed=9)
        mism = og != oo
        both = (og >= 0) & (oo >= 0)
        with np.errstate(invalid="ignore"):
            tie = mism & both & (np.abs(tg - to) <= TIE_RTOL * np.abs(to))
        hit_vs_miss = mism & ~both
        walk = og != ol
        same = ~mism & (oo >= 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            rel_t = np.abs(tg[same] - to[same]) / np.abs(to[same])
        row = {"scene": name, "rays": int(len(rays)), "id_mismatch": int(mism.sum()), "id_mismatch_rate": float(mism.mean()),
               "of_which_same_t_ties": int(tie.sum()), "of_which_hit_against_miss": int(hit_vs_miss.sum()),
               "walks_differ": int(walk.sum()), "t_bit_equal_where_ids_agree": float((tg[same] == to[same]).mean()),
               "max_rel_t_where_ids_agree": float(np.nanmax(rel_t)) if len(rel_t) else 0.0}
        print(json.dumps(row), flush=True)
        rows.append(row)
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump({"what": "rays aimed at rectangle borders / sphere touching points, device against CPU checker", "rows": rows}, f, indent=1)


if __name__ == "__main__":
    main()
